"""Cost of one node in a dependent chain of tiny kernels: hipGraph replay vs eager launches."""
import torch, time
dev=torch.device("cuda:0")
x=torch.zeros(64,device=dev)
def run(n):
    for _ in range(n): x.add_(1.0)
run(10); torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run(400)
g.replay(); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g.replay()
e1.record(); torch.cuda.synchronize()
print("graph chain of tiny kernels: %.2f us per node"%(e0.elapsed_time(e1)/4000*1000))
e0.record(); run(4000); e1.record(); torch.cuda.synchronize()
print("eager chain: %.2f us per launch"%(e0.elapsed_time(e1)/4000*1000))
