import numpy as np, sys, time, os, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from ceres_mono_orb_slam2_amd import synth, optimizer
from oracle import pyoracle as po
po.set_ba_threads(8)
def rel(a,b): return float(np.abs(np.asarray(a)-np.asarray(b)).max())
# C4 LocalBA full size
for nfix in (1,2):
  for dup in (True, False):
    g = synth.make_ba_graph(0, ncam=100, npts=10000, nobs=50000, n_fixed=nfix)
    a = (g["K4"], g["poses0"], g["cam_fixed"], np.ones(100,np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
    ab,poses,pts,er,s1,s2 = optimizer.local_bundle_adjustment(*a, duplicate_blocks=dup)
    t=time.time(); rc,oposes,opts,oer,o1,o2 = po.local_ba(*a, duplicate_blocks=dup); dt=time.time()-t
    print('C4 nfix',nfix,'dup',dup,'oracle %.1fs'%dt,'it',s1['iterations'],s2['iterations'],o1['iterations'],o2['iterations'],'term',s1['termination'],s2['termination'],o1['termination'],o2['termination'],
          'erase eq',np.array_equal(er,oer), int((er!=oer).sum()), 'cost1 rel', abs(s1['final_cost']-o1['final_cost'])/o1['final_cost'],'cost2 rel', abs(s2['final_cost']-o2['final_cost'])/o2['final_cost'],
          'pose',rel(poses,oposes),'pts',rel(pts,opts), flush=True)
# C5
for nfix,of in ((1,0.05),(2,0.05),(2,0.0)):
    g = synth.make_ba_graph(1000, ncam=500, npts=50000, nobs=250000, n_fixed=nfix, outlier_frac=of)
    n=len(g["obs_cam"]); w=np.asarray(g["obs_inv_sigma2"],np.float32).astype(np.float64); rb=np.ones(n,np.uint8)
    base=(g["K4"],g["poses0"],g["cam_fixed"],g["pts0"],g["obs_cam"],g["obs_pt"],g["obs_uv"],w,rb)
    for iters in (1,2,3,10):
        poses,pts,s = optimizer.bundle_adjustment(*base, iters)
        t=time.time(); oposes,opts,os_ = po.ba_solve(*base, iters); dt=time.time()-t
        print('C5 nfix',nfix,'of',of,'iters',iters,'oracle %.1fs'%dt,s['iterations'],os_['iterations'],s['successful_steps'],os_['successful_steps'],s['termination'],os_['termination'],
              'cost rel',abs(s['final_cost']-os_['final_cost'])/os_['final_cost'],'pose',rel(poses,oposes),'pts',rel(pts,opts),flush=True)
        if iters==3:
            # resync: one iteration from the oracle's 3-iteration state
            b2=(g["K4"],oposes,g["cam_fixed"],opts,g["obs_cam"],g["obs_pt"],g["obs_uv"],w,rb)
            p1,x1,s1=optimizer.bundle_adjustment(*b2,1); q1,y1,t1=po.ba_solve(*b2,1)
            print('   resync 1 iter from oracle state: cost rel',abs(s1['final_cost']-t1['final_cost'])/t1['final_cost'],'pose',rel(p1,q1),'pts',rel(x1,y1), s1['successful_steps'],t1['successful_steps'],flush=True)
