"""Sweeps the FIRST-mode chunk geometry (env knobs of the GRID engine) on rand10000."""
import os, sys, time, itertools, subprocess, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
if len(sys.argv) > 1 and sys.argv[1] == 'one':
    import numpy as np
    from tsp_optimization_amd import engine as E
    from helpers import load_instance
    ctx = E.Context(0)
    xy, wt = load_instance(os.environ.get('TSP_TUNE_INST', 'rand10000'))
    inst = E.Instance(ctx, xy, wt, 1)
    succ, obj, _ = inst.construct(E.GREEDY, np.array([0], dtype=np.int32))
    best = 1e9
    for rep in range(3):
        rc, s, o, st = inst.two_opt(succ[0], obj[0], mode=E.FIRST)
        best = min(best, st['device_ms'])
    print(json.dumps({'ms': best, 'steps': st['steps'], 'moves': st['moves'], 'cost': o, 'scanned': st['pairs_scanned']}))
else:
    for rpb, rmin, rmax, cnt in [(8,8,2048,1),(8,8,2048,0),(8,16,2048,1),(8,32,2048,1),(8,64,2048,1),(16,16,2048,1),(16,32,2048,1),(16,64,4096,1),(4,8,2048,1),(32,32,4096,1),(16,128,4096,1),(8,32,512,1),(8,32,8192,1)]:
        env = dict(os.environ, TSP_FIRST_ROWS_PER_BLOCK=str(rpb), TSP_FIRST_MIN_ROWS=str(rmin), TSP_FIRST_MAX_ROWS=str(rmax), TSP_COUNT_EVALS=str(cnt))
        out = subprocess.run([sys.executable, __file__, 'one'], env=env, capture_output=True, text=True)
        print(rpb, rmin, rmax, cnt, out.stdout.strip(), out.stderr.strip()[-200:])
