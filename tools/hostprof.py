import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
sys.argv = ['bench.py', '--steps', '6', '--warmup', '2', '--no-cpu-baseline']
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
