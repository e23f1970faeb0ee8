"""Can an RCCL all-reduce be recorded into a hipGraph next to this package's kernels and replayed?  One rank (world size 1: sums over
one rank are the identity, but the collective's kernel, its stream forks and the watchdog thread are all there).
  python tools/rccl_capture_probe.py        -> prints what worked"""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    flat = torch.zeros(1 << 20, device="cuda")
    src = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    dist.all_reduce(flat)                       # communicator up, outside any capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    res = {}
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            flat.copy_(src)
            ops.fill_zero(flat[:16])
            # async collective launched from a forked stream, waited for on the capture stream: the overlapped-bucket pattern
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                work = dist.all_reduce(flat, async_op=True)
            flat2 = flat[: 1 << 10] * 1.0       # (independent work on the capture stream)
            work.wait()
            torch.cuda.current_stream().wait_stream(side)
            out = flat * 2.0
        res["capture"] = "ok"
        for _ in range(3):
            src.add_(1.0)
            g.replay()
        torch.cuda.synchronize()
        want = (src * 2.0); want[:16] = 0
        res["replay_correct"] = bool(torch.equal(out, want))
        t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        res["replay_us"] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
    except Exception as e:      # noqa: BLE001
        res["capture"] = "FAILED: %s: %s" % (type(e).__name__, str(e)[:300])
    print("rccl_capture_probe:", res, flush=True)
    try:
        dist.barrier(); dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        print("teardown:", e)


if __name__ == "__main__":
    main()
