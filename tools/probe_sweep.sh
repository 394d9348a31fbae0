for p in 0 512 4096 65536 1073741824; do
  echo "== TSP_CLUSTER_PROBE=$p TSP_LDS_PROBE=$p"
  TSP_CLUSTER_PROBE=$p TSP_LDS_PROBE=$p timeout 300 python tools/shard_time.py 2>&1 | grep "engine auto"
  TSP_CLUSTER_PROBE=$p timeout 200 python tools/cluster_time.py quick 2>&1 | grep "FIRST CLUSTER  C=256"
done
