"""Developer GPU check (not part of the pytest suite): product HIP path vs the oracles."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import synth, tcv, orc, np_oracle as npo

def rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

print(tcv.lib().tcv_version().decode(), "devices", tcv.lib().tcv_device_count())
B = 4
batch = synth.make_windows(0, B)
w0 = synth.window_at(batch, 0)

# ---- factor evaluation parity
pr = w0["proj"]; n = len(pr["frame_i"])
params = np.concatenate([w0["pose"][pr["frame_i"]], w0["pose"][pr["frame_j"]], np.repeat(w0["ex_pose"][None], n, 0),
                         w0["lam"][pr["landmark"]][:, None]], 1)
pts = np.concatenate([pr["pts_i"], pr["pts_j"]], 1)
res, Js = tcv.eval_proj(pts, params, pr["sqrt_info"])
e = 0
for k in range(n):
    r, J = npo.proj_evaluate(params[k, 0:7], params[k, 7:14], params[k, 14:21], params[k, 21], pr["pts_i"][k], pr["pts_j"][k], pr["sqrt_info"])
    e = max(e, rel(res[k], r), *[rel(Js[b][k], J[b]) for b in range(4)])
print("proj eval max rel err", e)
ln = w0["line"]; nl = len(ln["frame"])
lparams = w0["pose"][ln["frame"]]
ld = np.concatenate([ln["pts_start"], ln["pts_end"], ln["abc"]], 1)
res, J = tcv.eval_line(ld, ln["K"], ln["Ric"], ln["Tic"], lparams)
e = 0
for k in range(nl):
    r, Jr = npo.line_evaluate(lparams[k], ln["pts_start"][k], ln["pts_end"][k], ln["abc"][k], ln["K"], ln["Ric"], ln["Tic"])
    e = max(e, rel(res[k], r), rel(J[k], Jr[0]))
print("line eval max rel err", e)
im = w0["imu"]; ni = len(im["frame_i"])
iparams = np.concatenate([w0["pose"][im["frame_i"]], w0["speedbias"][im["frame_i"]], w0["pose"][im["frame_j"]], w0["speedbias"][im["frame_j"]]], 1)
S_or = np.array([orc.imu_sqrt_info(im["covariance"][k]) for k in range(ni)])
res, Js, S = tcv.eval_imu(im, iparams, w0["G"])
print("imu sqrt_info dev vs oracle rel", rel(S, S_or))
res, Js, _ = tcv.eval_imu(im, iparams, w0["G"], sqrt_info=S_or)
e = 0
for k in range(ni):
    pre = dict(delta_p=im["delta_p"][k], delta_q=im["delta_q"][k], delta_v=im["delta_v"][k], lin_ba=im["lin_ba"][k], lin_bg=im["lin_bg"][k],
               sum_dt=float(im["sum_dt"][k]), jacobian=im["jacobian"][k], covariance=im["covariance"][k])
    r, J = npo.imu_evaluate(iparams[k, 0:7], iparams[k, 7:16], iparams[k, 16:23], iparams[k, 23:32], pre, w0["G"], sqrt_info=S_or[k])
    e = max(e, rel(res[k], r), *[rel(Js[b][k], J[b]) for b in range(4)])
print("imu eval (given sqrt_info) max rel err", e)
x = w0["pose"]; d = np.random.default_rng(0).normal(size=(x.shape[0], 6)) * 0.01
pp = tcv.pose_plus(x, d)
print("pose_plus err", max(rel(pp[k], npo.pose_plus(x[k], d[k])) for k in range(x.shape[0])))

# ---- solves
def run(wins, mfma, threads, iters=8, fixed=True, tag=""):
    W = [tcv.Window(w) for w in wins]
    b = tcv.Batch(W)
    o = tcv.default_options(iters, fixed, mfma, threads, True)
    t = time.time(); b.solve(o); b.synchronize(); dt = time.time() - t
    b.download_states()
    s = b.summaries()
    worst = 0
    for k, w in enumerate(wins):
        O = orc.Window(w)
        so = O.solve(iters, fixed)
        st = O.states(); sg = W[k].states()
        ni_ = min(so.num_iterations, s[k].num_iterations)
        ec = max(abs(s[k].cost[i] - so.cost[i]) / max(abs(so.cost[i]), 1e-300) for i in range(ni_))
        ex = max(rel(sg["pose"], st["pose"]), rel(sg["sb"], st["sb"]), rel(sg["ex"], st["ex"]), rel(sg["lam"], st["lam"]))
        fs = b.first_step(k)
        fo = np.array(so.first_delta[:so.n_local])
        ed = float(np.linalg.norm(fs - fo) / np.linalg.norm(fo)) if len(fs) == len(fo) else float("nan")
        em = max(abs(s[k].model_cost_change[i] - so.model_cost_change[i]) / max(abs(so.model_cost_change[i]), 1e-300) for i in range(1, ni_))
        print(f"  {tag} win{k}: its {s[k].num_iterations}/{so.num_iterations} term {s[k].termination}/{so.termination} "
              f"final {s[k].final_cost:.10g}/{so.final_cost:.10g} cost_rel {ec:.2e} dx1_rel {ed:.2e} model_rel {em:.2e} state_rel {ex:.2e} "
              f"cases {[s[k].dogleg_case[i] for i in range(ni_)]} / {[so.dogleg_case[i] for i in range(ni_)]}")
        worst = max(worst, ex)
    print(f" {tag} mfma={mfma} threads={threads}: kernel+sync {dt*1e3:.2f} ms, stats {b.stats()}, plan {b.plan_stats()}, worst state rel {worst:.2e}")

wins = [synth.window_at(batch, k) for k in range(B)]
for mf in (0, 1):
    for th in (256, 512):
        run(wins, mf, th, tag="noprior")
# prior from the oracle's marginalisation of the pre-window
pre = synth.make_windows(0, B, frame_shift=-1)
wp = []
for k in range(B):
    pw = synth.window_at(pre, k)
    O = orc.Window(pw); O.solve(8, True)
    prior, dbg = O.marginalize_old()
    w = dict(wins[k]); w["prior"] = prior
    wp.append(w)
    if k == 0: print("prior n", prior["n"], "m", prior["m"], "blocks", prior["blocks"])
run(wp, 1, 256, tag="prior")
run(wp, 0, 256, tag="prior")
run(wp, 1, 256, iters=30, fixed=False, tag="prior-converge")
# throughput glance
Bb = 512
big = synth.make_windows(0, Bb)
W = [tcv.Window(synth.window_at(big, k)) for k in range(Bb)]
b = tcv.Batch(W)
for mf, th in ((1, 256), (1, 512), (0, 256)):
    o = tcv.default_options(8, True, mf, th)
    b.solve(o); b.synchronize()
    t = time.time(); b.solve(o); b.synchronize(); dt = time.time() - t
    print(f"B={Bb} mfma={mf} th={th}: {dt*1e3:.2f} ms wall, event {b.stats()['solve_ms']:.2f} ms -> {Bb/dt:.0f} solves/s, {Bb*8/dt:.0f} it/s; plans {b.plan_stats()}")
