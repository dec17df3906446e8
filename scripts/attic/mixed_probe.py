import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
p = synth.scene_params(); sc = synth.make_scene(N, 2, seed=3, p=p)
X0 = sc["X0"].astype(np.float32).astype(np.float64); S0 = np.triu(sc["S0"]).astype(np.float32).astype(np.float64)
res = {}
for st in (srukf.STORAGE_F32, srukf.STORAGE_F32_MIXED):
    f = srukf.Filter(N, p); f.set_storage(st); f.set_state(X0, S0)
    f.predict_motion(sc["odo"][0], sc["odo"][1]); f.predict_measurement(); f.update(sc["z"][0], sc["matched"][0])
    X, S = f.get_state(); res[st] = (X, S.T @ S)
P1, P2 = res[srukf.STORAGE_F32][1], res[srukf.STORAGE_F32_MIXED][1]
D = np.abs(P1 - P2); n = 6 * N + 4
print("max |dP|", D.max(), "at", np.unravel_index(D.argmax(), D.shape), "P there", P1[np.unravel_index(D.argmax(), D.shape)])
print("LL", D[:n-4,:n-4].max(), "LR", D[:n-4,n-4:].max(), "RR", D[n-4:,n-4:].max(), "|P| LL", np.abs(P1[:n-4,:n-4]).max(), "RR", np.abs(P1[n-4:,n-4:]).max())
print("dX", np.abs(res[1][0]-res[2][0]).max())
rel = D / (np.sqrt(np.outer(np.diag(P1), np.diag(P1))) + 1e-30)
print("max rel (to sqrt(PiiPjj))", rel.max(), np.unravel_index(rel.argmax(), rel.shape))
