import numpy as np, torch, sys
sys.path.insert(0,'.')
from xmris_amd import device as dev
for n in (640,1280,2560,5120):
    for dt in ('complex64','complex128'):
        for nb in (1,4,7):
            rng=np.random.default_rng(1)
            x=(rng.standard_normal((nb,n))+1j*rng.standard_normal((nb,n))).astype(dt)
            ref=np.fft.fft(x.astype(np.complex128),axis=1,norm='ortho')
            got=dev.fft(dev.to_device(x),1).cpu().numpy()
            err=np.abs(got-ref)
            bad=np.argwhere(err>1e-4*np.abs(ref).max())
            print(n,dt,nb,'maxerr',err.max(),'nbad',len(bad), 'rows',sorted(set(bad[:,0].tolist())), 'cols', bad[:12,1].tolist())
