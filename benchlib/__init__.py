"""bench.py's parts: workloads, CPU baseline, PMC orchestration, sweeps, group drivers, the compact line."""
