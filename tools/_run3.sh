set -u
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r2c/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"
tail -8 gpurun_out/r2c/pytest_gpu.log
( time timeout 900 python bench.py ) > gpurun_out/r2c/bench_default.json 2> gpurun_out/r2c/bench_default.err; echo "bench rc=$?"
tail -3 gpurun_out/r2c/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2c/bench_default.json").read().strip().splitlines()[-1])
def show(x, name):
    print(name, "value %.4g %s  ms/step %.3f" % (x["value"], x["unit"], x["ms_per_step"]), "roofline", {k: (round(v,4) if isinstance(v,float) else v) for k,v in x["roofline"].items() if k in ("kernel","achieved","frac","kernel_ms","traffic")})
    for k in ("launch_mode","profiled_pass","stage_ms_per_step","launch_gap_ms_per_step","sign_iterations_per_signature","cpu_baseline","reference_published","end_to_end_host_fed","verdict_gather"):
        if k in x:
            v=x[k]
            if isinstance(v,dict): v={a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a not in ("note","sample","source")}
            print("   ",k,v)
show(d,"verify65")
for n,x in d["also"].items(): show(x,n)
print(d["library_stats"])
PY
