for i in 1 2; do
echo "== W6 for P=4 (tree build)"; python profiles/r04/probe_kernels.py 5
echo "== W5 for P=4"; PYTHONPATH=profiles/r04/ab_w5 python - <<'PY'
import sys, runpy
sys.argv = ["probe_kernels.py", "5"]
sys.path.insert(0, "profiles/r04/ab_w5")
src = open("profiles/r04/probe_kernels.py").read().replace("sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))", "")
exec(compile(src, "probe", "exec"))
PY
done
