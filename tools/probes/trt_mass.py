import numpy as np, sys
sys.path.insert(0,'.')
from latticeboltzmannsimulations_amd import CavitySolver, ghia
for RT,dtype,arith,kernel in (("TRT",np.float32,"strict","auto"),("TRT",np.float32,"fast","stream"),("TRT",np.float32,"fast","generic"),("TRT",np.float64,"strict","auto"),("TRT",np.float64,"fast","auto"),("SRT",np.float32,"fast","stream"),("MRT",np.float32,"fast","stream")):
    with CavitySolver(256,256,1000.0,RT=RT,dtype=dtype,arith=arith,kernel=kernel) as s:
        prev=None; quiet=0
        for _ in range(400):
            s.step(3000)
            u,rho=s.get_fields(out_dtype=np.float64)
            m=float(np.mean(u))
            if prev is not None and abs(m-prev)/0.08<1e-8:
                quiet+=1
                if quiet>5: break
            prev=m
        _,_,fin=s.get_fields(want_fin=True,out_dtype=np.float64)
        ex,ey=ghia.profile_errors(u,1000,0.08)
        print(RT,np.dtype(dtype).name,arith,kernel,"steps",s.steps_done,"mass drift %.4f"%((fin.sum()-65536)/65536),"ghia %.4f %.4f"%(ex,ey),flush=True)
