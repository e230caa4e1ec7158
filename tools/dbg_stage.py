import sys, time, os
sys.path.insert(0, os.getcwd())
def log(*a):
    print(f"[{time.time()-T0:7.2f}]", *a, flush=True)
T0=time.time()
import numpy as np
import torch
log("torch imported", torch.cuda.is_available())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem
N=int(sys.argv[1]) if len(sys.argv)>1 else 200
mp = problem.make_map(N, density=0.02, seed=0); log("map", mp.occupancy.mean())
pts = mp.obstacle_points(); log("pts", pts.shape)
ctx = gtop.GtopContext(device=0); log("ctx")
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution); log("init_sdf_map", ctx.grid)
ctx.update_sdf_map(pts); log("update_sdf_map done")
d = ctx.get_sdf(); log("get_sdf", d.shape, d.min(), d.max())
batch = problem.make_trajectories(1024, 6, mp, seed=1); log("traj")
dev=torch.device("cuda:0")
x=torch.tensor(batch.x,device=dev); Df=torch.tensor(batch.Df.reshape(-1,18),device=dev); T=torch.tensor(batch.T,device=dev)
c,g=ctx.eval_device(x,Df,T); torch.cuda.synchronize(); log("eval ok", float(c[0]))
if len(sys.argv)>2:
    gph=torch.cuda.CUDAGraph()
    cost=torch.zeros(4,1024,dtype=torch.float64,device=dev); grad=torch.zeros_like(x)
    with torch.cuda.graph(gph):
        for s in range(4): ctx.eval_device(x,Df,T,cost[s],grad)
    log("captured")
    gph.replay(); torch.cuda.synchronize(); log("replayed", float(cost[3,0]))
