import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as g
pkg=g.load_package(); import oracle_lib as orc
from gpu_util import dev, ptr
ctx=pkg.HipContext.on_torch_stream(0)
rng=np.random.default_rng(0)
print("== gemm NN/NT checks")
for (m,n,k) in [(9,9,24),(44,44,56),(24,44,32),(16,44,16),(8,44,16),(100,100,100),(200,130,70),(300,300,64),(129,257,33)]:
    for kn in (0,1):
        A=rng.standard_normal((m,k)); B=rng.standard_normal((k,n)) if kn else rng.standard_normal((n,k)); Cm=rng.standard_normal((m,n))
        lda,ldb,ldc=k+2,(n+2 if kn else k+2),n+4
        dA=torch.zeros((m,lda),dtype=torch.float64,device='cuda'); dA[:,:k]=dev(A)
        dB=torch.zeros((B.shape[0],ldb),dtype=torch.float64,device='cuda'); dB[:,:B.shape[1]]=dev(B)
        dC=torch.zeros((m,ldc),dtype=torch.float64,device='cuda'); dC[:,:n]=dev(Cm)
        ctx.gemm_minus(m,n,k,ptr(dA),lda,ptr(dB),ldb,kn,ptr(dC),ldc,0); ctx.sync()
        want=Cm-(A@B if kn else A@B.T)
        err=np.abs(dC.cpu().numpy()[:,:n]-want).max()
        print(m,n,k,'kn' if kn else 'nt','err',err, 'pad',float(dC[:,n:].abs().max()))
print("== LU sweep")
for n in (33,34,40,48,56,63,64,65,72,80,100):
    a=np.random.default_rng(n).standard_normal((n,n))
    lda=n+2
    d_a=torch.zeros((n,lda),dtype=torch.float64,device='cuda'); d_a[:,:n]=dev(a)
    d_p=torch.zeros(n,dtype=torch.int32,device='cuda')
    sg=ctx.lu_decomp(n,ptr(d_a),lda,ptr(d_p))
    lu=d_a.cpu().numpy()[:,:n]; perm=d_p.cpu().numpy()
    lu_o,perm_o,sg_o=orc.lu_decomp(a)
    L=np.tril(lu,-1)+np.eye(n); U=np.triu(lu)
    print(n,'perm_eq',np.array_equal(perm,perm_o.astype(np.int64)),'resid',np.abs(L@U-a[perm]).max(),'maxL',np.abs(np.tril(lu,-1)).max(), 'first diff', int(np.argmax(perm!=perm_o.astype(np.int64))))
