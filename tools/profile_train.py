import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["train_sort.py"] + sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("train_sort", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "train_sort.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
pr = cProfile.Profile(); pr.enable(); mod.main(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
