import ctypes as C, time, sys
sys.path.insert(0,'.')
import fade_amd
ctx=fade_amd.Context(device=0)
L=ctx._L
for mb in (16,128,128,128,128,512):
    p=C.c_void_p(); t=time.perf_counter(); rc=L.fadehip_host_alloc(ctx._h, mb<<20, C.byref(p)); t1=time.perf_counter()
    C.memset(p, 1, mb<<20); t2=time.perf_counter()
    print("hipHostMalloc %d MB: %.1f ms, first touch %.1f ms"%(mb,(t1-t)*1e3,(t2-t1)*1e3))
import numpy as np
a=np.empty(128<<20,np.uint8); t=time.perf_counter(); a[:]=1; print("malloc+touch 128MB pageable %.1f ms"%((time.perf_counter()-t)*1e3))
