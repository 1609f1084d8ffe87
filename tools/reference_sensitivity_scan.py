#!/usr/bin/env python3
"""Is there ANY fixture on which the reference's own chained result is stable to 1e-4?  (build container only.)

Runs the UNMODIFIED reference optimiser (reconstruct/optimizer.py via oracle/ref_shim.py) with an ANALYTIC decoder -- a torch module that evaluates
the shape family's SDF in closed form, so there is no fitted network and no fit error at all -- on synthetic objects of two shape families
(ellipsoid: curved everywhere; rounded box: flat faces), three sizes and two initialisations (the bench's 0.25 m / 5 deg noise, and 2 cm / 0.5 deg),
and re-runs each with every input element moved to an adjacent float32 (4 seeded draws).  Prints the largest movement of the reference's own final
pose / code.  Output of the round-3 run: profiles/r03_reference_sensitivity_scan.md -- no configuration is below 3e-4 on the pose; the sensitivity is
the algorithm's (threshold flips + amplification through near-solid samples), not a property of a decoder fixture.
"""
import sys, json, os, tempfile, contextlib, io, time, itertools
import numpy as np, torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim
from dsp_slam_amd import synth
ref_shim.install()
from reconstruct.optimizer import Optimizer
from reconstruct.utils import get_configs
HALF = torch.tensor([0.38,0.28,0.80])
FAMILY='ellipsoid'
class Analytic(nn.Module):
    def forward(self, x):
        code3 = x[..., :3]; p = x[..., -3:]
        r = HALF*(1+0.2*torch.tanh(code3))
        if FAMILY=='ellipsoid':
            k0 = torch.sqrt(((p/r)**2).sum(-1)+1e-12); k1 = torch.sqrt(((p/(r*r))**2).sum(-1)+1e-12)
            f = k0*(k0-1)/k1
        else:
            q = p.abs()-r
            f = torch.clamp(q,min=0).norm(dim=-1) + torch.clamp(q.max(-1).values, max=0) - 0.08
        return torch.tanh(f).unsqueeze(-1)
def ell_sdf(p, code3, half=None):
    p=np.asarray(p,np.float64); r=synth.shape_half_extents(code3, half)
    k0=np.sqrt(((p/r)**2).sum(-1)+1e-12); k1=np.sqrt(((p/(r*r))**2).sum(-1)+1e-12)
    return k0*(k0-1)/k1
box_sdf = synth.rounded_box_sdf
def cfg(iters):
    d={"data_type":"KITTI","DeepSDF_DIR":"/nonexistent","voxels_dim":32,"optimizer":{"code_len":64,"num_depth_samples":50,"cut_off_threshold":0.01,
 "joint_optim":dict(k1=1.0,k2=100.0,k3=0.25,k4=1e7,b1=0.2,b2=0.025,num_iterations=iters,learning_rate=1.0,scale_damping=1.0),"pose_only_optim":{"num_iterations":5,"learning_rate":1.0}}}
    f=tempfile.NamedTemporaryFile('w',suffix='.json',delete=False); json.dump(d,f); f.close()
    return get_configs(f.name)
def jiggle(a,rng):
    up=rng.integers(0,2,size=a.shape).astype(bool)
    return np.where(up,np.nextafter(a,np.float32(np.inf)),np.nextafter(a,np.float32(-np.inf))).astype(np.float32)
def trial(family, seed, M, B, t_noise, yaw, iters=10, ndraw=4):
    global FAMILY
    FAMILY=family
    synth.rounded_box_sdf = ell_sdf if family=='ellipsoid' else box_sdf
    opt=Optimizer(Analytic(), cfg(iters))
    obj=synth.make_object(seed, n_surface=M, n_background=B, t_noise=t_noise, yaw_noise_deg=yaw)
    def run(pts,rays,depth):
        with contextlib.redirect_stdout(io.StringIO()):
            return opt.reconstruct_object(obj['t_cam_obj_init'].copy(), pts.copy(), rays.copy(), depth.copy())
    base=run(obj['pts'],obj['rays'],obj['depth'])
    if not base.is_good: return None
    rng=np.random.default_rng(5); dts=[];dcs=[]
    for i in range(ndraw):
        r=run(jiggle(obj['pts'],rng),jiggle(obj['rays'],rng),jiggle(obj['depth'],rng))
        dts.append(float(np.abs(r.t_cam_obj-base.t_cam_obj).max())); dcs.append(float(np.abs(r.code-base.code).max()))
    gt=obj['t_cam_obj_gt']
    return dict(terr=float(np.linalg.norm(base.t_cam_obj[:3,3]-gt[:3,3])), dT=max(dts), dC=max(dcs), code=base.code[:3].round(3).tolist(), gt=obj['code_gt'][:3].round(3).tolist())
if __name__=='__main__':
    for fam in ('ellipsoid','box'):
        for (M,B) in ((2000,500),(2000,0),(500,0)):
            for (tn,yaw) in ((0.25,5.0),(0.02,0.5)):
                r=trial(fam,1,M,B,tn,yaw)
                print(fam,M,B,tn,yaw,r, flush=True)
