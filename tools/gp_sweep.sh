# developer sweep of the group_points LDS kernel's launch shape (run on the GPU box from the repo root)
mkdir -p gpurun_out/r3/gp; export TMPDIR=/tmp
for cfg in "0 0 1" "0 0 0" "0 256 0" "0 512 0"; do
  set -- $cfg
  rm -rf gpurun_out/r3/gp/*
  PWCLO_GP_THREADS=$1 PWCLO_GP_TARGET=$2 PWCLO_GP_NT=$3 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3/gp -o gp -- python tools/group_roofline.py run > /dev/null 2>&1
  f=$(find gpurun_out/r3/gp -name "*kernel_trace.csv" | head -1)
  echo "== threads=$1 target=$2 nt=$3"
  python tools/group_roofline.py report $f | tail -11 | awk '{print $1,$2,$3,$4, $6, $7, $10}'
done
rm -rf gpurun_out/r3/gp
