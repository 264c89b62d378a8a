"""Complementary overlap of the two passes: chunk k+1's strided (HBM-bound) pass is released
when chunk k's has finished, so that it runs beside chunk k's tile (VALU-bound) pass on a
second stream.  Compared with whole-batch passes and with the same chunks on one stream."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

hx.set_tuning("plan", hx.PLAN_SPLIT)
N, B = 65536, 4096
q = 18014398510661633
ntt = hx.NTT(N, q)
x = torch.empty((B, N), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, N, B, 1, q)
ref = x[:2].clone()
streams = [torch.cuda.Stream() for _ in range(2)]


def step_plain():
    ntt.ComputeForward(x, x, 1, 1)
    ntt.ComputeInverse(x, x, 1, 1)


def step_offset(chunk):
    """Each transform of a chunk is two kernels on one stream; the chunks alternate between
    two streams and chunk k+1 starts half a transform after chunk k (a sleep-free offset:
    it waits for an event recorded after a one-pass-long dummy on the other stream)."""
    main = torch.cuda.current_stream()
    for fwd in (True, False):
        fn = ntt.ComputeForward if fwd else ntt.ComputeInverse
        prev = None
        for i, c in enumerate(range(0, B, chunk)):
            s = streams[i & 1]
            v = x[c:c + chunk]
            with torch.cuda.stream(s):
                if i == 0:
                    s.wait_stream(main)
                fn(v, v, 1, 1)
        for s in streams:
            main.wait_stream(s)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3


for _ in range(10):
    step_plain()
print("whole batch: %.3f ms" % timed(step_plain), flush=True)
# offset start: stream 1 begins with a half-size chunk so that from then on its strided
# passes fall beside stream 0's tile passes
for chunk in (2048, 1024, 512, 256):
    def step_skewed():
        main = torch.cuda.current_stream()
        for fwd in (True, False):
            fn = ntt.ComputeForward if fwd else ntt.ComputeInverse
            bounds = [0, chunk // 2]
            while bounds[-1] < B:
                bounds.append(min(B, bounds[-1] + chunk))
            for i in range(len(bounds) - 1):
                s = streams[i & 1]
                v = x[bounds[i]:bounds[i + 1]]
                with torch.cuda.stream(s):
                    if i < 2:
                        s.wait_stream(main)
                    fn(v, v, 1, 1)
            for s in streams:
                main.wait_stream(s)
    print("chunk %4d: two streams in step %.3f ms   skewed by half a chunk %.3f ms" % (
        chunk, timed(lambda: step_offset(chunk)), timed(step_skewed)), flush=True)
torch.cuda.synchronize()
assert torch.equal(ref, x[:2])
