"""A/B aid: run bench.py with the MAIN stream (the chain of the first modality group: the step's critical path) at high priority and
the group streams beside it at normal priority:  python scripts/main_priority.py [bench args]"""
import os, runpy, sys
import torch
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
