"""Where a slab of the chained split-bf16 forward kernel spends its time: shader-clock stamps of every wave around every chunk barrier (probe build
tools/build_chain_split_stamps.sh; BG_LIB=tools/probe/libbg_split_stamps.so python tools/chain_split_stamps.py [critic_wgs actor_wgs]).  Per network, the
median over all waves of: cycles waiting at the top of each chunk (wait + barrier), cycles issuing the copies, cycles of the chunk's body, next to the MFMA
cycles of the chunk (tiles x 2 k-steps x 9 x 32), and the shader clock (cycles / 100 MHz wall ticks).  `pair`: both networks side by side on two streams."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from booster_gym_amd import _lib
import test_gpu_mlp_chain_split as T
lib = _lib.load(); st = _lib.current_stream_ptr()
lib.bg_probe_read_split_stamps.restype = C.c_int; lib.bg_probe_read_split_stamps.argtypes = [C.c_void_p, C.c_size_t]
wgs = {"critic": int(sys.argv[1]) if len(sys.argv) > 1 else 160, "actor": int(sys.argv[2]) if len(sys.argv) > 2 else 96}
nets = {"critic": (102400, (64, 256, 256, 128), 61), "actor": (98304, (64, 256, 128, 128), 47)}
cases = {k: T._case(M, dims, seed=3, k_real=kr, wgs=wgs[k]) for k, (M, dims, kr) in nets.items()}


def report(name, tag):
    """stamps: [0] kernel start, [1 + 2 c] arrival at the top of chunk c of the LAST slab of the workgroup, [2 + 2 c] released by its barrier,
    [1 + 2 C] kernel end (behind the last slab's tail), [62] / [63] 100 MHz wall clock at start / end"""
    M, dims, _ = nets[name]
    K0, N1, N2, N3 = dims
    buf = np.zeros(2 * 256 * 4 * 64, dtype=np.int64)
    assert lib.bg_probe_read_split_stamps(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(2, 256, 4, 64)[int(N2 == 256)][: wgs[name]]
    chunks = [(K0 // 32, N1), (N1 // 32, N2), (N2 // 32, N3)]
    Cn = sum(c for c, _ in chunks)
    mf = [n // 32 * 2 * 9 * 32 for c, n in chunks for _ in range(c)]
    arrive, released = t[:, :, 1 : 1 + 2 * Cn : 2], t[:, :, 2 : 2 + 2 * Cn : 2]
    end = t[:, :, 1 + 2 * Cn]
    nxt = np.concatenate((arrive[:, :, 1:], end[:, :, None]), axis=2)
    med = lambda a: np.median(a.reshape(-1, a.shape[-1]), axis=0)
    wait, body = med(released - arrive), med(nxt - released)
    slabs = -(-M // 128)
    per_wg = -(-slabs // wgs[name])
    tot = np.median(end - t[:, :, 0])
    ghz = np.median((end - t[:, :, 0]) / np.maximum(1, t[:, :, 63] - t[:, :, 62])) * 0.1
    print(f"{name} [{tag}]: kernel {tot:.0f} cycles = {tot / ghz / 1e3:.1f} us at {ghz:.2f} GHz for {per_wg} slabs = {tot / per_wg:.0f} cycles per slab; MFMA {sum(mf)} per slab")
    print("   last slab, chunk: wait / body (MFMA)   " + "  ".join(f"{w:.0f}/{b:.0f}({f})" for w, b, f in zip(wait, body, mf)), flush=True)
    print(f"   sums: wait {wait.sum():.0f}  body {body.sum():.0f} (the last chunk's body includes the kernel's tail)", flush=True)


for name in nets:
    d = cases[name][0]
    for _ in range(5):
        _lib.check(lib.bg_mlp_chain_forward_split(C.addressof(d), 1, st))
    torch.cuda.synchronize()
    report(name, f"alone, {wgs[name]} workgroups")
side = torch.cuda.Stream()
for _ in range(5):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _lib.check(lib.bg_mlp_chain_forward_split(C.addressof(cases["critic"][0]), 1, _lib.current_stream_ptr()))
    _lib.check(lib.bg_mlp_chain_forward_split(C.addressof(cases["actor"][0]), 1, st))
    torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for name in nets:
    report(name, "pair on two streams")
