# k_psi_so adaptive capped at 128 registers (4 waves per SIMD, mesh capacity 224 so that 16 waves
# fit the LDS) against the tree's 160 registers / 3 waves per SIMD; config 4, kernels one after
# the other (probe_kernels.py) and the whole loop (probe_ab-style timing)
for i in 1 2; do
echo "== tree (160 VGPRs, 3 waves/SIMD)"; python profiles/r04/probe_kernels.py 4
echo "== 128 VGPRs, 4 waves/SIMD"; python - <<'PY'
import sys
sys.argv = ["probe_kernels.py", "4"]
sys.path.insert(0, "profiles/r04/ab_so4")
src = open("profiles/r04/probe_kernels.py").read().replace("sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))", "")
exec(compile(src, "probe", "exec"))
PY
done
