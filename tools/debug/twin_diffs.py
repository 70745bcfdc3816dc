import sys; sys.path.insert(0,'tests')
import numpy as np, abi_driver as A
for kind,ncell,npk,its in (("stromgren",16,30000,3),("stromgren_diffuse",16,30000,3),("lexington",16,30000,5)):
    e=A.engine(ncell); c=A.twin(ncell)
    got,(xH,xHe,T)=A.run_benchmark(e,kind,ncell,npk,its)
    ref,(xH0,xHe0,T0)=A.run_benchmark(c,kind,ncell,npk,its)
    for k,((tw,tc,JH,JHe,hH),(tw0,tc0,JH0,JHe0,hH0)) in enumerate(zip(got,ref)):
        print(kind,k,tc-tc0, np.abs(JH-JH0).max()/JH0.max(), np.abs(JHe-JHe0).max()/max(JHe0.max(),1e-300), np.abs(hH-hH0).max()/max(np.abs(hH0).max(),1e-300))
    print(kind,'xH rel',np.max(np.abs(xH-xH0)/np.maximum(xH0,1e-300)),'T rel',np.max(np.abs(T-T0)/np.maximum(T0,1e-300)))
