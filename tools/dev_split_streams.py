"""Developer tool: does splitting the benchmark batch into sub-batches on separate HIP streams (the marginalisation of one part overlapping the
solve tail of another) raise the throughput?   python tools/dev_split_streams.py [B] [parts]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import torch, synth, tcv, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
opts = tcv.default_options(8, True, True, 256)
def run(batches, streams, steps=20, warm=3):
    def step():
        for b, s in zip(batches, streams): b.solve(opts, s)
        for b, s in zip(batches, streams): b.gauge_fix(s)
        for b, s in zip(batches, streams): b.marginalize(s)
    def step_seq():
        for b, s in zip(batches, streams): b.solve(opts, s); b.gauge_fix(s); b.marginalize(s)
    out = []
    for fn in (step, step_seq):
        for _ in range(warm): fn()
        for b in batches: b.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): fn()
        for b in batches: b.synchronize()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    return out
for parts in ([1, 2, 4] if len(sys.argv) < 3 else [int(sys.argv[2])]):
    per = B // parts
    batches = [bench.build_batches(tcv, synth, 100000 + k * per, per) for k in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    ms = run([b[0] for b in batches], [s.cuda_stream for s in streams])
    print("B = %d in %d part(s) of %d on %d stream(s): %.3f ms per step interleaved (%.1f K solves/s), %.3f ms part by part (%.1f K)" % (B, parts, per, parts, ms[0], B / ms[0], ms[1], B / ms[1]), flush=True)
    del batches
