import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as g
pkg=g.load_package()
ctx=pkg.HipContext.on_torch_stream(0)
n,dim,m=8192,2,10_000_000
f64=torch.float64
d_x=torch.empty((n,dim),dtype=f64,device='cuda'); d_y=torch.empty((m,dim),dtype=f64,device='cuda'); d_s=torch.empty(m,dtype=f64,device='cuda')
ctx.synth_unit(0xC0FFEE01,0,0.0,1.0,d_x.data_ptr(),n*dim); ctx.synth_unit(0xC0FFEE02,0,0.02,0.96,d_y.data_ptr(),m*dim)
d_f=torch.sin(3*d_x[:,0])+torch.sin(6*d_x[:,1]); eps=2.0*n**0.5
d_phi=torch.empty((n,n),dtype=f64,device='cuda'); d_w=torch.empty(n,dtype=f64,device='cuda')
xsum0=float(d_x.sum())
for it in range(8):
    d_w.copy_(d_f)
    st,route=ctx.rbf_solve(0,eps,d_x.data_ptr(),n,dim,dim,d_phi.data_ptr(),n,d_w.data_ptr())
    print('iter',it,'solve st',st,'route',route, flush=True)
    if st!=0:
        st2,info=ctx.cholesky_decomp1(n,d_phi.data_ptr(),n); print('  direct decomp on current phi: ',st2,info)
        ctx.rbf_fill(0,eps,d_x.data_ptr(),n,dim,dim,d_phi.data_ptr(),n); torch.cuda.synchronize()
        print('  refilled phi diag',float(d_phi.diagonal().min()),float(d_phi.diagonal().max()),'sym',bool((d_phi==d_phi.T).all()))
        st2,info=ctx.cholesky_decomp1(n,d_phi.data_ptr(),n); print('  decomp after refill: ',st2,info)
    ctx.rbf_eval(0,eps,d_x.data_ptr(),n,dim,dim,d_w.data_ptr(),d_y.data_ptr(),m,dim,d_s.data_ptr())
