import sys, numpy as np
sys.path.insert(0,'.')
from oracle import ea_oracle as eo
from edge_alignment_amd import synth, capi
q=synth.quat_from_axis_angle([1, 2, 3], np.deg2rad(1.0))
pr=synth.make_problem(120, 160, 2000, 40, 6, 130.0, 130.0, 79.5, 59.5, planted_q=q, planted_t=(0.01, -0.005, 0.02), normalize=True)
X=pr['xyz'].copy()
X[10] = [0.1, 0.1, 0.005]; X[700] = [0.0, 0.0, -0.009]; X[20] = [5.0, 0.0, 1.0]; X[21] = [-5.0, -4.0, 1.0]; X[22] = [0.0, 9.0, 1.0]; X[23] = [0.61, 0.455, 1.0]; X[24] = [1e6, -1e6, 1.0]; X[25] = [0.3, 0.2, -2.0]
O=eo.OracleProblem(pr['grid'],*pr['K'])
P=capi.Problem(*pr['K'], dtype=capi.EA_F64); P.set_points(X); P.set_dt_grid(pr['grid'])
qq = np.array([0.7, 0.05, -0.04, 0.02]) * 1.3
a=O.eval(X,qq,np.zeros(3),eo.JAC_JET,materialize=True)
r,J=P.eval_points(qq,np.zeros(3),corrected=False)
d=np.abs(J-a['raw_J']); d[np.isnan(d)]=0; d[24]=0
for i in np.argsort(d.max(axis=1))[-4:]:
    print(i, X[i], 'gpu', J[i], 'oracle', a['raw_J'][i], 'r', r[i], a['raw_r'][i])
