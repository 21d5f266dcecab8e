import numpy as np, time, xxhash, os
from concurrent.futures import ThreadPoolExecutor
a=np.random.rand(90,196,64*3)
mv=memoryview(a.reshape(-1).view(np.uint8))
for _ in range(3):
    t=time.perf_counter(); d=xxhash.xxh3_128(mv).digest(); t1=time.perf_counter()-t
    print("single", round(t1*1e3,3),"ms", round(a.nbytes/t1/1e9,1),"GB/s")
for n in (2,4,8,16):
    ex=ThreadPoolExecutor(n)
    def h(i):
        sz=len(mv); lo=i*sz//n; hi=(i+1)*sz//n
        return xxhash.xxh3_128(mv[lo:hi]).digest()
    list(ex.map(h, range(n)))
    for _ in range(3):
        t=time.perf_counter(); ds=list(ex.map(h, range(n))); t2=time.perf_counter()-t
    print(n,"threads", round(t2*1e3,3),"ms", round(a.nbytes/t2/1e9,1),"GB/s")
print(os.cpu_count(), len(os.sched_getaffinity(0)))
b=a.copy()
t=time.perf_counter(); e=np.array_equal(a,b); print("array_equal", round((time.perf_counter()-t)*1e3,3))
