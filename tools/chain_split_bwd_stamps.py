"""Where a slab of the chained split-bf16 backward kernel spends its time: shader-clock stamps of every wave around every chunk barrier (probe build
tools/build_chain_split_bwd_stamps.sh; BG_LIB=tools/probe/libbg_bwd_stamps.so python tools/chain_split_bwd_stamps.py [critic_wgs actor_wgs]).  Per network, the median
over all waves of the LAST slab of every workgroup: cycles waiting at the top of each chunk (counted wait + barrier), cycles in the compiler's own wait for the
chunk's loads, cycles of the chunk's body next to its MFMA cycles (k-steps x 9 x 32), and the shader clock."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from booster_gym_amd import _lib
import test_gpu_mlp_chain_split_bwd as T
lib = _lib.load(); st = _lib.current_stream_ptr()
lib.bg_probe_read_bwd_stamps.restype = C.c_int; lib.bg_probe_read_bwd_stamps.argtypes = [C.c_void_p, C.c_size_t]
wgs = {"critic": int(sys.argv[1]) if len(sys.argv) > 1 else 160, "actor": int(sys.argv[2]) if len(sys.argv) > 2 else 96}
nets = {"critic": (98304, (256, 256, 128)), "actor": (98304, (256, 128, 128))}
cases = {k: T._case(M, dims, 3, wgs[k]) for k, (M, dims) in nets.items()}


def report(name, tag):
    M, (N1, N2, N3) = nets[name]
    buf = np.zeros(2 * 256 * 4 * 64, dtype=np.int64)
    assert lib.bg_probe_read_bwd_stamps(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(2, 256, 4, 64)[int(N2 == 256)][: wgs[name]]
    TA, TB = N2 // 32, N1 // 32
    Cn = TA + TB
    mf = [N3 // 16 * 9 * 32] * TA + [N2 // 16 * 9 * 32] * TB
    arrive, released, touched = t[:, :, 1 : 1 + 3 * Cn : 3], t[:, :, 2 : 2 + 3 * Cn : 3], t[:, :, 3 : 3 + 3 * Cn : 3]
    end = t[:, :, 1 + 3 * Cn]
    nxt = np.concatenate((arrive[:, :, 1:], end[:, :, None]), axis=2)
    med = lambda a: np.median(a.reshape(-1, a.shape[-1]), axis=0)
    wait, touch, body = med(released - arrive), med(touched - released), med(nxt - touched)
    per_wg = -(-(M // 128) // wgs[name])
    tot = np.median(end - t[:, :, 0])
    ghz = np.median((end - t[:, :, 0]) / np.maximum(1, t[:, :, 63] - t[:, :, 62])) * 0.1
    print(f"{name} [{tag}]: kernel {tot:.0f} cycles = {tot / ghz / 1e3:.1f} us at {ghz:.2f} GHz for {per_wg} slabs = {tot / per_wg:.0f} cycles per slab; MFMA {sum(mf)} per slab")
    print("   last slab, chunk: barrier wait / loads' wait / body (MFMA)   " + "  ".join(f"{w:.0f}/{u:.0f}/{b:.0f}({f})" for w, u, b, f in zip(wait, touch, body, mf)), flush=True)
    ka, kb = np.diff(t[:, :, 40:48], axis=2), np.diff(t[:, :, 48:62], axis=2)
    print("   layer A chunk 1, cycles per k-step (288 MFMA): " + " ".join(f"{v:.0f}" for v in med(ka)) + ";  layer B chunk 2: " + " ".join(f"{v:.0f}" for v in med(kb)))
    print(f"   sums: barrier wait {wait.sum():.0f}  loads' wait {touch.sum():.0f}  body {body.sum():.0f} (the first body holds the slab's last k-step and the last the kernel's tail)", flush=True)


for name in nets:
    d = cases[name][0]
    fin = _lib.ReduceProblem()
    for _ in range(5):
        _lib.check(lib.bg_mlp_chain_backward_split(C.addressof(d), 1, fin, st))
    torch.cuda.synchronize()
    report(name, f"alone, {wgs[name]} workgroups")
