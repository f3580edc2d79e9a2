python -m pytest tests/test_gpu_device_arrays.py -x -q -p no:cacheprovider 2>&1 | tail -4
python - <<'PY'
import time, numpy as np, opticommpy_amd as oa
x = (np.random.default_rng(0).normal(size=(1<<20,2)) + 0j)
d = oa.to_device(x); d.get()
t0=time.perf_counter(); d = oa.to_device(x); t1=time.perf_counter(); y = d.get(); t2=time.perf_counter()
print(f"to_device 32 MiB {1e3*(t1-t0):.2f} ms ({x.nbytes/(t1-t0)/1e9:.1f} GB/s), get {1e3*(t2-t1):.2f} ms ({x.nbytes/(t2-t1)/1e9:.1f} GB/s)", np.array_equal(x,y))
PY
