import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem
from oracle import oracle
mp = problem.make_map((60, 50, 30), density=0.03, seed=11)
ctx = gtop.GtopContext(0); ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution); ctx.update_sdf_map(mp.obstacle_points()); ctx.set_params()
sdf = oracle.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size); sdf.build_from_occupancy(mp.occupancy)
dev = torch.device("cuda:0")
for m in (2, 4, 6, 7, 10):
    for B in (23, 3101):
        b = problem.make_trajectories(B, m, mp, seed=5 + m)
        for td, tol in ((torch.float64, 1e-9), (torch.float32, 2e-4)):
            x, Df, T = (torch.tensor(a, dtype=td, device=dev) for a in (b.x, b.Df.reshape(-1, 18), b.T))
            ctx.set_launch_geometry(0, 10)
            c, g = ctx.eval_device(x, Df, T); torch.cuda.synchronize()
            c_ref, g_ref, _ = oracle.eval_batch(b.T, b.Df, b.x, sdf, oracle.make_params(), nthreads=8)
            rc = np.max(np.abs(c.double().cpu().numpy() - c_ref) / np.abs(c_ref))
            rg = np.max(np.max(np.abs(g.double().cpu().numpy() - g_ref), axis=1) / np.max(np.abs(g_ref), axis=1))
            print(m, B, td, rc, rg, "OK" if rc <= tol and rg <= tol else "BAD", flush=True)
