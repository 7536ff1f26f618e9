"""Does the VALU-bound ExpandMask of the NEXT signing round fit under the HBM-bound w = invNTT(A_hat o NTT(y)) of the current one?
Runs mldsa_verify_arith (the same kernel family as sign_w, A_hat re-read per op: HBM-bound) and mldsa_expand_mask for 65 536
ops of ML-DSA-65 back to back on one stream and side by side on two, and prints the times.  (tools/ubench_overlap.hip asked the
same question for ExpandMask || SampleInBall: no gain, both VALU.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fips204_amd.hotpath import HotPath

hp = HotPath(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
k, l = 6, 5
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randint(0, 8380417, (n, k, l, 256), dtype=torch.int32, device="cuda", generator=g)
z = torch.randint(-(1 << 19) + 1, 1 << 19, (n, l, 256), dtype=torch.int32, device="cuda", generator=g)
c = torch.zeros((n, 256), dtype=torch.int32, device="cuda"); c[:, :49] = 1
t1 = torch.randint(0, 8380417, (n, k, 256), dtype=torch.int32, device="cuda", generator=g)
rho = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
kappa = torch.zeros(n, dtype=torch.int16, device="cuda")
s1 = torch.cuda.Stream()
cands = [torch.cuda.Stream() for _ in range(4)]  # HIP streams share four hardware queues: try several partners
s2 = cands[0]
w_out = torch.empty((n, k, 256), dtype=torch.int32, device="cuda")

def arith():
    return hp.verify_arith(65, a, z, c, t1, out=w_out)
def mask(m=n):
    return hp.expand_mask(65, rho[:m], kappa[:m])

def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

def serial(m):
    with torch.cuda.stream(s1):
        arith(); mask(m)
def both(m):
    with torch.cuda.stream(s1):
        arith()
    with torch.cuda.stream(s2):
        mask(m)
def only_a():
    with torch.cuda.stream(s1): arith()
print("verify_arith65 %d ops alone: %.1f us" % (n, timed(only_a)))
for m in (n, n * 4 // 5, n // 2):
  for s2 in cands:
    def only_m():
        with torch.cuda.stream(s2): mask(m)
    def both(m):
        with torch.cuda.stream(s1):
            arith()
        with torch.cuda.stream(s2):
            mask(m)
    print("expand_mask65 %6d ops alone: %.1f us | arith then mask on one stream: %.1f us | side by side on two streams: %.1f us"
          % (m, timed(only_m), timed(lambda: serial(m)), timed(lambda: both(m))))
