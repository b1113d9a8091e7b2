"""CPU emulation (numpy) of the fp16 x 2 split products of csrc/meanshift_h2.h against a plain fp32
GEMM, both measured against fp64: S = X X^T, the kernel values and one mean-shift step on
clustered unit vectors at four noise levels / bandwidths.  python tools/h2_accuracy.py"""
import numpy as np


def unit(a):
    return a / np.linalg.norm(a, axis=1, keepdims=True)


def split16(a, scale):
    a = (a * scale).astype(np.float32)
    h = a.astype(np.float16)
    m = (a - h.astype(np.float32)).astype(np.float16)
    return h.astype(np.float64), m.astype(np.float64)


def mm3(A, Bt, sa, sb):
    """A Bt^T from the three significant piece products (piece products are exact in fp32; the
    accumulation is emulated in fp64 and rounded once)."""
    ah, am = split16(A, sa)
    bh, bm = split16(Bt, sb)
    return (ah @ bh.T + ah @ bm.T + am @ bh.T) / (sa * sb)


def main():
    rng = np.random.default_rng(0)
    N, D = 3000, 128
    code = unit(rng.standard_normal((12, D)))
    lab = rng.integers(0, 12, N)

    def err(a, ref):
        return np.abs(a - ref).max() / np.abs(ref).max()

    for noise, b in [(0.02, 0.15), (0.3, 0.8), (0.003, 0.02), (1.0, 1.2)]:
        X = unit(code[lab] + noise * rng.standard_normal((N, D))).astype(np.float32)
        X64 = X.astype(np.float64)
        hl = 0.5 / (b * b)

        def kern(S):
            return np.exp(np.clip(-(2 - 2 * S) * hl, -75, 75))
        S64 = X64 @ X64.T
        K64 = kern(S64)
        U64 = (K64 @ X64) / K64.sum(1, keepdims=True)
        S32 = X @ X.T
        K32 = kern(S32.astype(np.float64)).astype(np.float32)
        U32 = (K32 @ X) / K32.sum(1, keepdims=True)
        S16 = mm3(X, X, 2.0 ** 12, 2.0 ** 12).astype(np.float32)
        K16 = kern(S16.astype(np.float64)).astype(np.float32)
        U16 = mm3(K16, X.T.copy(), 2.0 ** 14, 2.0 ** 12).astype(np.float32) / K16.sum(1, keepdims=True)
        print("noise %-5g b %-4g | S: fp32 %.2e fp16x2 %.2e | K: fp32 %.2e fp16x2 %.2e | step: fp32 %.2e fp16x2 %.2e"
              % (noise, b, err(S32, S64), err(S16, S64), np.abs(K32 - K64).max(), np.abs(K16 - K64).max(),
                 err(U32, U64), err(U16, U64)))


if __name__ == "__main__":
    main()
