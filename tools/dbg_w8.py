import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
torch.cuda.is_available()
import isingmontecarlo_amd as im, _lattices as lat
L=int(sys.argv[1]); W=int(sys.argv[2]); R=8
edges = lat.cubic_periodic(L); nsite=L**3
J = np.random.default_rng(1).choice([-1.0,1.0], size=(R,len(edges)))
beta=4.0
cap = 1 << int(np.ceil(np.log2(2.0*beta*(len(edges)*1.3+nsite*1.2)+4*nsite)))
g = im.QmcIsingGraph(edges, 1.0, 0.1, nsite, 5, nreplicas=R, capacity=cap, waves_per_replica=W, couplings=J)
print(g.launch_info())
for it in range(12):
    try:
        g.run(2, beta)
    except Exception as e:
        print("it", it, "ERR", e); break
    print(it, "n", g.get_n()[:3], "cut", g.get_cutoff()[:3], "verify", g.verify().all())
