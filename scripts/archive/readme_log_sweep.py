"""Can the reference's recorded solve (standalone/README.md:26-71: initial cost 8.743202, final 0.5418352, 30 steps,
YPR (-0.32, 1.52, 2.50) deg, t (-0.01, 0.00, -0.05)) be reproduced from the bundled frames?  CPU only (the oracle and
the numpy restatement of the pre-processing).  Part 1: A = frame 1 (44457 points, stride 30 = 1482 blocks) against
B = 2..5 with the three distance-transform producers.  Part 2: for B = 5 (the only frame that lands on the logged pose)
every combination of blur, channel order, edge threshold, median filter and distance-transform mask: initial cost.
Output committed as profiles/r02_readme_log_sweep.txt."""
import os, sys, itertools, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import preprocess_np as pp, ea_oracle as eo
from edge_alignment_amd import synth
G=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests','golden','rgbd'); K=(525.0,525.0,319.5,239.5)
imgs={b: pp.load_rgb_as_bgr(os.path.join(G,'rgb_%d.png'%b)) for b in range(1,6)}
deps={b: pp.load_depth_u16(os.path.join(G,'depth_%d.png'%b)) for b in range(1,6)}
aX,_=pp.get_aX(imgs[1],deps[1],*K)
X=aX[:3,::30].T.copy()

def ypr(q):
    R=synth.quat_to_R(q)
    return (np.degrees(np.arctan2(R[1,0],R[0,0])), np.degrees(np.arctan2(-R[2,0],np.hypot(R[2,1],R[2,2]))), np.degrees(np.arctan2(R[2,1],R[2,2])))
def dt_lap(img, median=True, thr=35):
    es = pp.edge_strength(img)
    Bm = np.where(es > thr, 0, 255).astype(np.uint8)
    if median: Bm = pp.median_blur3_u8(Bm)
    return pp.normalize_minmax_f32(pp.distance_transform_l2_3(Bm))
print('# part 1: A = 1, stride 30, CauchyLoss(1), identity start, LM defaults; log: init 8.743202 final 0.5418352 30 it ypr (-0.32,1.52,2.50) t (-0.01,0.00,-0.05)')
for bb in (2,3,4,5):
    for name,dt in (('laplacian+median (shipped)',dt_lap(imgs[bb],True)),('laplacian, no median',dt_lap(imgs[bb],False)),('canny 30/90 (get_distance_transform2)',pp.get_distance_transform2(imgs[bb]))):
        P=eo.OracleProblem(pp.grid_view_of_image(dt),*K)
        e=P.eval(X,[1,0,0,0],[0,0,0]); q,t,s=P.solve(X,[1,0,0,0],[0,0,0]); y=ypr(q)
        print('B%d %-38s init %.6f final %.7f it %d succ %d %s ypr (%.2f,%.2f,%.2f) t (%.3f,%.3f,%.3f)'%(bb,name,e['cost'],s['final_cost'],s['num_iterations'],s['num_successful_steps'],s['why'],y[0],y[1],y[2],t[0],t[1],t[2]), flush=True)
print('# part 2: B = 5, initial cost by pre-processing variant (log: 8.743202)')
b=5
img=imgs[b]
def es_of(img, blur=True, rgb_order='bgr'):
    im = img if rgb_order=='bgr' else img[:,:,::-1].copy()
    bl = pp.gaussian_blur3_u8(im) if blur else im
    g = pp.rgb2gray_u8(bl)
    return pp.laplacian3_abs_u8(g)
t0=time.time()
for blur, order in itertools.product((True,False),('bgr','rgb')):
    es = es_of(img, blur, order)
    for thr in (20,25,30,35,40,45,50,60):
        for med in (True,False):
            Bm = np.where(es > thr, 0, 255).astype(np.uint8)
            if med: Bm = pp.median_blur3_u8(Bm)
            for kind in ('l2_3','precise'):
                d = pp.distance_transform_l2_3(Bm) if kind=='l2_3' else pp.distance_transform_precise(Bm)
                dt = pp.normalize_minmax_f32(d)
                P=eo.OracleProblem(pp.grid_view_of_image(dt),*K)
                e=P.eval(X,[1,0,0,0],[0,0,0])
                flag = ' <==' if abs(e['cost']-8.743202)<0.05 else ''
                print('blur %d order %s thr %d med %d %s init %.6f%s'%(blur,order,thr,med,kind,e['cost'],flag), flush=True)
