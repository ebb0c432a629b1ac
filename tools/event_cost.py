"""GPU-side cost of a cross-stream hand-over on the main queue: N x [kernel; record; side waits; side kernel] against
N x [kernel], with torch events and raw HIP events (hipEventDisableSystemFence / release-to-device).  Run under
rocprofv3 --kernel-trace; tools/event_cost_gaps.py prints the gaps between consecutive main-queue kernels per phase."""
import ctypes
import torch

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 26, device=dev)     # 256 MB: ~100 us per add_
y = torch.zeros(1 << 24, device=dev)
main = torch.cuda.Stream()
side = torch.cuda.Stream()
N = 60


def run(kind):
    evs = []
    if kind.startswith("raw"):
        flags = {"raw_default": 0x2, "raw_nofence": 0x2 | 0x20000000, "raw_release_dev": 0x2 | 0x40000000}[kind]
        for _ in range(N):
            e = ctypes.c_void_p()
            assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0
            evs.append(e)
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        x.mul_(1.0)                     # phase marker (mul: a different kernel name)
        for i in range(N):
            x.add_(1.0)
            if kind == "none":
                continue
            if kind == "torch":
                e = torch.cuda.Event()
                e.record()
                side.wait_event(e)
            elif kind == "torch_wait_only":      # the main stream WAITS for an (already complete) side event
                pass
            else:
                assert hip.hipEventRecord(evs[i], ctypes.c_void_p(main.cuda_stream)) == 0
                assert hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), evs[i], 0) == 0
            if kind != "torch_wait_only":
                with torch.cuda.stream(side):
                    y.add_(1.0)
            else:
                with torch.cuda.stream(side):
                    e = torch.cuda.Event()
                    e.record()
                main.wait_event(e)
    torch.cuda.synchronize()
    for e in evs:
        hip.hipEventDestroy(e)


for kind in ("none", "torch", "raw_default", "raw_nofence", "raw_release_dev", "torch_wait_only"):
    run(kind)
    print(kind)
