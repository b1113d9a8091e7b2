"""Drop-in ``src`` package: the reference's module paths and call signatures
(Hippogriff/parsenet-codebase, ``src/*.py``) backed by the MI355X-native implementation in
``parsenet_codebase_amd``.  Only the hot path is provided; everything here
requires tensors on the GPU — there is no CPU fallback."""
