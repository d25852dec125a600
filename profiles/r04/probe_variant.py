"""Run-average kernel durations of a coupled config with ANOTHER build of the engine (a package
directory under profiles/r04/, git-ignored): timing experiments whose results may be wrong on
purpose.  usage: python profiles/r04/probe_variant.py <dir under profiles/r04> 3 4 5"""
import sys, os
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, sys.argv[1]))
sys.argv = [sys.argv[0]] + sys.argv[2:]
src = open(os.path.join(here, "probe_kernels.py")).read().replace(
    "sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))", "")
exec(compile(src, "probe_kernels", "exec"))
