import sys, os
sys.path.insert(0, os.getcwd())
import torch
from asr_study_amd import ops
from tools.gpu_microbench import timeit
dev='cuda:0'
T,n_pad,H=999,32,256
rows=T*n_pad
x=torch.randn(rows,2*H,device=dev); W=torch.randn(2*H,8*H,device=dev)*0.05
z=torch.empty(rows,8*H,device=dev); b=torch.randn(8*H,device=dev)
m=(torch.rand(2,n_pad,2*H,device=dev)>0.2).float()/0.8
fl=2.0*rows*4*H*2*H
t=timeit(lambda: ops.gemm(x,W,z,rows,8*H,2*H,bias=b)); print('fwd full N=2048: %.3f ms %.1f TF/s'%(t,2*fl/t/1e9))
t=timeit(lambda: ops.gemm(x,W,z,rows,4*H,2*H,ldb=8*H,ldc=8*H,bias=b[:4*H])); print('fwd half nomask: %.3f ms %.1f TF/s'%(t,fl/t/1e9))
t=timeit(lambda: ops.gemm(x,W,z,rows,4*H,2*H,ldb=8*H,ldc=8*H,bias=b[:4*H],a_scale=m[0],a_scale_period=n_pad)); print('fwd half mask: %.3f ms %.1f TF/s'%(t,fl/t/1e9))
dz=torch.randn(rows,8*H,device=dev); dx=torch.empty(rows,2*H,device=dev)
t=timeit(lambda: ops.gemm(dz,W,dx,rows,2*H,8*H,trans_b=True)); print('dx full K=2048: %.3f ms %.1f TF/s'%(t,2*fl/t/1e9))
t=timeit(lambda: ops.gemm(dz,W,dx,rows,2*H,4*H,trans_b=True,lda=8*H,ldb=8*H)); print('dx half nomask: %.3f ms %.1f TF/s'%(t,fl/t/1e9))
t=timeit(lambda: ops.gemm(dz,W,dx,rows,2*H,4*H,trans_b=True,lda=8*H,ldb=8*H,beta=1.0,c_scale=m[0],c_scale_period=n_pad)); print('dx half mask beta1: %.3f ms %.1f TF/s'%(t,fl/t/1e9))
dW=torch.empty(2*H,8*H,device=dev)
t=timeit(lambda: ops.gemm(x,dz,dW,2*H,8*H,rows,trans_a=True,split_k='auto')); print('dW full: %.3f ms %.1f TF/s'%(t,2*fl/t/1e9))
t=timeit(lambda: ops.gemm(x,dz,dW,2*H,4*H,rows,trans_a=True,ldb=8*H,ldc=8*H,split_k='auto')); print('dW half nomask: %.3f ms %.1f TF/s'%(t,fl/t/1e9))
t=timeit(lambda: ops.gemm(x,dz,dW,2*H,4*H,rows,trans_a=True,ldb=8*H,ldc=8*H,split_k='auto',a_scale=m[0],a_scale_period=n_pad)); print('dW half mask: %.3f ms %.1f TF/s'%(t,fl/t/1e9))
