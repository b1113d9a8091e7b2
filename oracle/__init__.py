"""TEST INFRASTRUCTURE ONLY.

``oracle/`` is a CPU restatement of the reference's hot path (plain C for the index /
selection arithmetic, torch-CPU / numpy for the rest).  It exists to *check* the HIP
product and to serve as bench.py's ``cpu_baseline``.  It must never be imported from
``parsenet_codebase_amd`` or ``src`` — the product has no CPU path.
"""
