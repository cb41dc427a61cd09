# group_points roofline evidence (run on the GPU box from the repo root): device-side durations from rocprofv3's kernel trace
out=${1:-gpurun_out/r3/group_points_roofline.txt}
mkdir -p gpurun_out/gp_tmp $(dirname $out); export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gp_tmp -o gp -- python tools/group_roofline.py run > /dev/null 2>&1
python tools/group_roofline.py report "$(find gpurun_out/gp_tmp -name '*kernel_trace.csv' | head -1)" > $out
echo >> $out; echo "rocprofv3 --stats of the same run (all shapes of a kernel name pooled):" >> $out
grep -i "group_points" "$(find gpurun_out/gp_tmp -name '*kernel_stats.csv' | head -1)" >> $out
rm -rf gpurun_out/gp_tmp
cat $out
