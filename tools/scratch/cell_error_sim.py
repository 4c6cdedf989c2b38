import numpy as np
rng = np.random.default_rng(1)
N = 400000
K = 5
def unit32(v):
    v = v.astype(np.float32)
    n2 = np.zeros(len(v), np.float32)
    for k in range(K): n2 = np.float32(n2 + v[:,k]*v[:,k])
    inv = (np.float32(1)/np.sqrt(n2)).astype(np.float32)
    return (v*inv[:,None]).astype(np.float32)
x = rng.normal(size=(N,K)); a = rng.normal(size=(N,K))
# f32 inputs (raw) as the reference sees them
xr = x.astype(np.float32); ar = a.astype(np.float32)
# exact cos in f64 of the f32 inputs
def cos64(x,a):
    x=x.astype(np.float64); a=a.astype(np.float64)
    return (x*a).sum(1)/np.sqrt((x*x).sum(1)*(a*a).sum(1))
exact = 1-cos64(xr,ar)
# reference f32: dot products sequential mul+add, sqrt(na*nb), div
def dot32(p,q):
    s=np.zeros(len(p),np.float32)
    for k in range(K): s=np.float32(s+np.float32(p[:,k]*q[:,k]))
    return s
ref = np.float32(1)-np.float32(dot32(xr,ar)/np.sqrt(np.float32(dot32(xr,xr)*dot32(ar,ar))))
# kernel: unit vectors in f32 then products
xu = unit32(xr); au = unit32(ar)
def fma_chain(xu,au):
    d=np.ones(len(xu),np.float64)
    for k in range(K): d = np.float32(d - xu[:,k].astype(np.float64)*au[:,k].astype(np.float64)).astype(np.float64)
    return d
reg = fma_chain(xu,au)
def trunc_bf16(v): return (v.view(np.uint32)&np.uint32(0xffff0000)).view(np.float32)
def rtn_bf16(v):
    u=v.view(np.uint32).astype(np.uint64); u=(u+0x7fff+((u>>16)&1))&0xffff0000
    return u.astype(np.uint32).view(np.float32)
def split3(v,f):
    p0=f(v); r=np.float32(v-p0); p1=f(r); p2=np.float32(r-p1); assert np.all(f(p2)==p2)
    return [p0.astype(np.float64),p1.astype(np.float64),p2.astype(np.float64)]
def bf3(xu,au,fx,fa):
    tot=np.ones(len(xu),np.float64)
    for k in range(K):
        X=split3(np.ascontiguousarray(xu[:,k]),fx); A=split3(np.ascontiguousarray(au[:,k]),fa)
        for i in range(3):
            for j in range(3):
                if i+j<=2: tot-=X[i]*A[j]
    return np.float32(tot).astype(np.float64)  # one rounding at the end (idealised accumulate)
def f16x2(xu,au):
    def tr16(v): return (v.view(np.uint32)&np.uint32(0xffffe000)).view(np.float32)
    tot=np.ones(len(xu),np.float64)
    g=np.float32(1+1.1e-7)
    for k in range(K):
        xv=np.ascontiguousarray(xu[:,k]); x0=tr16(xv); x1=tr16(np.float32(xv-x0))
        av=np.float32(au[:,k]*g); a0=av.astype(np.float16).astype(np.float32); a1=np.float32(av-a0).astype(np.float16).astype(np.float32)
        tot-= x0.astype(np.float64)*a0+x1.astype(np.float64)*a0+x0.astype(np.float64)*a1
    return np.float32(tot).astype(np.float64)
for name,v in [("ref f32",ref.astype(np.float64)),("reg fma",reg),("bf3 trunc/rtn",bf3(xu,au,trunc_bf16,rtn_bf16)),("bf3 rtn/rtn",bf3(xu,au,rtn_bf16,rtn_bf16)),("f16x2",f16x2(xu,au))]:
    e=v-exact
    print("%-14s mean %+.3e rms %.3e max %.3e"%(name,e.mean(),np.sqrt((e*e).mean()),np.abs(e).max()))
