"""What does a cross-stream hand-over cost the stream that records the event?  A chain of 200 dependent ~60 us kernels on one stream (the host runs ahead), timed with
(a) nothing between them, (b) a torch.cuda.Event recorded after every kernel (what Stream.wait_stream / record_event do), (c) a HIP event created with
hipEventDisableTiming | hipEventDisableSystemFence, (d) ... | hipEventReleaseToDevice recorded after every kernel; and the same with a second stream
waiting on every event and launching a small kernel of its own (the fork pattern of the rollout and of the update).
    python tools/event_cost_probe.py -> gpurun_out/event_cost_probe.json"""
import ctypes, json, os, sys
import torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
dev = "cuda:0"
x = torch.zeros(1 << 25, device=dev); y = torch.zeros(4096, device=dev)  # x.add_: ~60 us of GPU work per launch, so that the host runs ahead of the GPU
main, side = torch.cuda.Stream(), torch.cuda.Stream()
K = 200
def hip_events(flags):
    evs = []
    for _ in range(K):
        e = ctypes.c_void_p(); assert hip.hipEventCreateWithFlags(ctypes.byref(e), flags) == 0; evs.append(e)
    return evs
def run(mode, evs=None, fork=False):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(main):
        e0.record()
        for k in range(K):
            x.add_(1.0)
            if mode == "torch":
                ev = torch.cuda.Event(); ev.record(main)
                if fork: side.wait_event(ev)
            elif mode == "hip":
                assert hip.hipEventRecord(evs[k], ctypes.c_void_p(main.cuda_stream)) == 0
                if fork: assert hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), evs[k], 0) == 0
            if fork:
                with torch.cuda.stream(side): y.add_(1.0)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3
out = {}
flagsets = {"hip_disable_timing": 0x2, "hip_no_system_fence": 0x2 | 0x20000000, "hip_release_to_device": 0x2 | 0x40000000}
for fork in (False, True):
    tag = "fork" if fork else "record_only"
    for rep in range(3):
        out.setdefault(f"{tag}/none", []).append(round(run("none", fork=False), 2))
        out.setdefault(f"{tag}/torch_event", []).append(round(run("torch", fork=fork), 2))
        for name, fl in flagsets.items():
            out.setdefault(f"{tag}/{name}", []).append(round(run("hip", hip_events(fl), fork=fork), 2))
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"us_per_kernel_of_a_200_kernel_dependent_chain": out}, open("gpurun_out/event_cost_probe.json", "w"), indent=1)
