import sys; sys.path.insert(0,'/root/repo/tc-viml_amd')
import numpy as np, replay, ate
S=8; F=300; feats=100
streams = [replay.simulate_stream_euroc(seq, F, start_s=0.5 + 0.35 * k, seed=k, max_features=feats, max_lines=20, associate=(k % 2 == 0)) for seq in replay.EUROC_SEQUENCES for k in range(S)]
idx=int(sys.argv[1])
st=streams[idx]
nat=replay.run_many_native([st],8)[0]
py=replay.run_many([st], replay.HipBackend(), 8)[0]
gt=st["gt_p"][replay.WINDOW_SIZE:]
en=np.linalg.norm(nat["p"]-gt[:len(nat["p"])],axis=1); ep=np.linalg.norm(py["p"]-gt[:len(py["p"])],axis=1)
for k in range(0,len(en),20): print(k, "native %.3f python %.3f"%(en[k],ep[k]), nat["log"][k]["n_line"], py["log"][k]["n_line"], nat["log"][k]["n_proj"], "cost %.0f"%nat["log"][k]["final_cost"])
