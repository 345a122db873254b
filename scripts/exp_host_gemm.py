import torch, time
dev=torch.device("cuda:0")
x=torch.randn(2708,1433,device=dev); W=torch.randn(64,1433,device=dev); b=torch.randn(64,device=dev)
Wt=W.t().contiguous(); out=torch.empty(2708,64,device=dev)
def t(f,n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    return (t1-t0)/n*1e6,(t2-t0)/n*1e6
with torch.no_grad():
    print("linear      host %.1f total %.1f us"%t(lambda: torch.nn.functional.linear(x,W,b)))
    print("mm(W.t())   host %.1f total %.1f us"%t(lambda: torch.mm(x,W.t())))
    print("mm(Wt)      host %.1f total %.1f us"%t(lambda: torch.mm(x,Wt)))
    print("mm out=     host %.1f total %.1f us"%t(lambda: torch.mm(x,Wt,out=out)))
    print("addmm       host %.1f total %.1f us"%t(lambda: torch.addmm(b,x,Wt)))
    print("matmul      host %.1f total %.1f us"%t(lambda: x@Wt))
