"""Where the host time of a lock-step replay goes (cProfile, cumulative): window management (Python) vs the C-ABI calls."""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
import replay
streams = [replay.simulate_stream_euroc(s, 80, start_s=0.5, max_features=60, max_lines=8) for s in replay.EUROC_SEQUENCES]
be = replay.HipBackend()
replay.run_many(streams[:1], be, num_iterations=8)      # warm-up
pr = cProfile.Profile(); pr.enable()
replay.run_many(streams, be, num_iterations=8)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
