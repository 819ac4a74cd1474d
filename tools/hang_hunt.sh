# fresh-process stress of the C++ node: where does the rare hang sit?
cd $GRAFT_REPO_ROOT; make -C examples >/dev/null 2>&1
fails=0
for i in $(seq 1 380); do
  P=lanes; [ $((i % 2)) -eq 0 ] && P=scan
  MPPI_NODE_TRACE=/tmp/progress.bin timeout 5 ./build/mppi_node --task park --samples 2048 --horizon 50 --callbacks 30 --seed 5 --storage f64 --tick-path $P > /dev/null 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "run $i path=$P rc=$rc progress=$(od -An -i -N4 /tmp/progress.bin)"; fi
  [ $fails -ge 3 ] && break
done
echo "$fails failures in $i runs"
