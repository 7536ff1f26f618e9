"""Does the download of sub-batch 1's signatures overlap with the signing of sub-batch 2?  Device-resident signing of two
32 768-op halves on one stream, the D2H of the first half on another stream (event-ordered), timed with and without it."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from fips204_amd.hotpath import HotPath
hp = HotPath(0)
wl = bench.WholeOp(hp, 65, "sign", 65536, 0)
ml = wl.ml
n, h = 65536, 32768
sig = torch.empty((n, ml.SIG_LEN), dtype=torch.uint8, device="cuda")
st = torch.zeros(n, dtype=torch.int32, device="cuda")
host_sig = torch.empty((n, ml.SIG_LEN), dtype=torch.uint8, pin_memory=True)
comp, down = torch.cuda.Stream(), torch.cuda.Stream()
hp.set_option(9, 2)
def half(i):
    a, b = i * h, (i + 1) * h
    ml.sign_device(wl.sks, wl.msg_buf, wl.msg_off[a:], wl.rnd[a:b], sig[a:b], h, key_idx=wl.key_idx[a:b], status=st[a:b], wait=False)
def run(with_d2h, reps=5):
    for _ in range(2):
        with torch.cuda.stream(comp): half(0); half(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        with torch.cuda.stream(comp):
            half(0)
            e = torch.cuda.Event(); e.record(comp)
        if with_d2h:
            with torch.cuda.stream(down):
                down.wait_event(e)
                host_sig[:h].copy_(sig[:h], non_blocking=True)
        with torch.cuda.stream(comp):
            half(1)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
def d2h_only(reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        with torch.cuda.stream(down): host_sig[:h].copy_(sig[:h], non_blocking=True)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("two halves, no download      %.3f ms" % run(False))
print("D2H of one half alone         %.3f ms" % d2h_only())
print("two halves + D2H(half 1) beside half 2   %.3f ms" % run(True))
