"""Where a slab of the chained forward kernel spends its time: shader-clock stamps of every wave behind every chunk barrier (probe build of the library with
-DBG_CHAIN_PROBE_STAMPS, built by tools/build_chain_stamps.sh: BG_LIB=tools/probe/libbg_chain_stamps.so python tools/mlp_chain_stamps.py).  Prints, per network, the median over all waves of the
cycles between consecutive stamps next to the MFMA cycles of that chunk (N / 32 tiles x 16 k-steps x 64 cycles), for the first and a later round of slabs."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from booster_gym_amd import _lib
lib = _lib.load(); dev = "cuda:0"; st = _lib.current_stream_ptr(); p = _lib.ptr
lib.bg_probe_read_chain_stamps.restype = C.c_int; lib.bg_probe_read_chain_stamps.argtypes = [C.c_void_p, C.c_size_t]
for name, M, dims in (("actor", 98304, (64, 256, 128, 128)), ("critic", 102400, (64, 256, 256, 128))):
    K0, N1, N2, N3 = dims
    x = torch.randn(M, K0, device=dev)
    Ws = [torch.randn(n, k, device=dev) / k ** 0.5 for k, n in ((K0, N1), (N1, N2), (N2, N3))]
    bs = [torch.randn(n, device=dev) * 0.1 for n in (N1, N2, N3)]
    ys = [torch.empty(M, n, device=dev) for n in (N1, N2, N3)]
    d = _lib.MlpChain(M, K0, N1, N2, N3, 0, p(x), p(Ws[0]), p(bs[0]), p(Ws[1]), p(bs[1]), p(Ws[2]), p(bs[2]), p(ys[0]), p(ys[1]), p(ys[2]))
    for _ in range(5):
        _lib.check(lib.bg_mlp_chain_forward_group(C.addressof(d), 1, st))
    torch.cuda.synchronize()
    buf = np.zeros(2048 * 4 * 24, dtype=np.int64)
    assert lib.bg_probe_read_chain_stamps(buf.ctypes.data, buf.nbytes) == 0
    nsl = M // 128
    t = buf.reshape(2048, 4, 24)[: min(nsl, 2048)]
    chunks = [(K0 // 32, N1), (N1 // 32, N2), (N2 // 32, N3)]
    Cn = sum(c for c, _ in chunks)
    mf = [n // 32 * 16 * 64 for c, n in chunks for _ in range(c)]
    dt = np.diff(t[:, :, : Cn + 2], axis=2)  # [slab][wave][Cn + 1]: prologue, chunk 0 .. Cn-1 (the last up to the end of the slab)
    start = t[:, :, 0].min(axis=1)
    order = np.argsort(start)
    for label, sel in (("first round", order[:256]), ("second round", order[256:512])):
        med = np.median(dt[sel].reshape(-1, Cn + 1), axis=0)
        tot = np.median((t[sel][:, :, Cn + 1] - t[sel][:, :, 0]).reshape(-1))
        print(f"{name} {label}: slab {tot:.0f} cycles, MFMA {sum(mf)}; prologue {med[0]:.0f}; chunks (measured / MFMA): " +
              " ".join(f"{m:.0f}/{f}" for m, f in zip(med[1:], mf)), flush=True)
