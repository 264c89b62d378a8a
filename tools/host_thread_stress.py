"""Stress of the host-pointer entry points from many threads at once (each thread has its own stream, bounce buffer,
slots and completion flag): every result compared with the single-threaded one.  ctypes releases the GIL during the calls."""
import ctypes as C
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

THREADS = int(os.environ.get("THREADS", "32"))
CALLS = int(os.environ.get("CALLS", "400"))
shapes = [(4096, 49, 1), (8192, 54, 1), (8192, 54, 3), (16384, 54, 1), (16384, 49, 2), (65536, 54, 1), (32768, 60, 1),
          (131072, 54, 1), (2048, 28, 5)]
plans, refs, inputs = [], [], []
p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
for n, bits, batch in shapes:
    q = hx.GeneratePrimes(1, bits, True, n)[0]
    ntt = hx.NTT(n, q)
    x = np.random.default_rng(n + batch).integers(0, q, (batch, n), dtype=np.uint64)
    y = np.zeros_like(x)
    assert hx.lib.hexl_amd_ntt_forward_host(ntt._h, p(y), p(x), batch, 1, 1) == 0
    plans.append(ntt)
    inputs.append(x)
    refs.append(y)
errors = []


def worker(tid):
    rng = np.random.default_rng(tid)
    for k in range(CALLS):
        i = int(rng.integers(0, len(shapes)))
        n, bits, batch = shapes[i]
        buf = inputs[i].copy()
        if hx.lib.hexl_amd_ntt_forward_host(plans[i]._h, p(buf), p(buf), batch, 1, 1) != 0:
            errors.append((tid, k, "rc"))
            return
        if not np.array_equal(buf, refs[i]):
            errors.append((tid, k, shapes[i], "forward differs"))
            return
        if hx.lib.hexl_amd_ntt_inverse_host(plans[i]._h, p(buf), p(buf), batch, 1, 1) != 0 or \
                not np.array_equal(buf, inputs[i]):
            errors.append((tid, k, shapes[i], "inverse differs"))
            return


ts = [threading.Thread(target=worker, args=(t,)) for t in range(THREADS)]
for t in ts:
    t.start()
for t in ts:
    t.join()
print(f"{THREADS} threads x {CALLS} forward+inverse host calls over {len(shapes)} shapes: "
      f"{'all results identical to the single-threaded ones' if not errors else errors[:5]}; "
      f"host_polls {hx.get_counter('host_polls')}, host_poll_timeouts {hx.get_counter('host_poll_timeouts')}")
sys.exit(1 if errors else 0)
