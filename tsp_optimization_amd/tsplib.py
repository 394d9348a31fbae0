"""TSPLIB reader for the Python side of the package (NODE_COORD_SECTION instances only, like the reference's
parse_instance, src/utility.c:351-453: keyword lines split on " :\\n\\t\\r", coordinates through atof, node ids
1-based).  The C host mirror has its own (tsp_host.c:parse_instance); this one serves bench.py and tools."""
import re

import numpy as np

WEIGHT_TYPES = {"EUC_2D": 0, "MAX_2D": 1, "MAN_2D": 2, "CEIL_2D": 3, "GEO": 4, "ATT": 5}   # include/utility.h:45-52


def parse(path):
    """-> (xy float64 [n, 2], weight type number; unknown types map to EUC_2D like src/distutil.c:90-91)"""
    n, wt, xy, in_coords = -1, 0, None, False
    with open(path) as f:
        for line in f:
            tok = [t for t in re.split(r"[ :\n\t\r]+", line) if t]
            if not tok:
                continue
            # keyword lines are recognised wherever they stand (the reference's cascade runs on every line,
            # src/utility.c:371-449: test/data/shuffled_prop_att48.tsp carries header fields after the coordinates)
            if tok[0] == "DIMENSION":
                n = int(tok[1])
            elif tok[0] == "EDGE_WEIGHT_TYPE":
                wt = WEIGHT_TYPES.get(tok[1], 0)
            elif tok[0] == "NODE_COORD_SECTION":
                if n <= 0:
                    raise ValueError("%s: NODE_COORD_SECTION before DIMENSION" % path)
                xy = np.zeros((n, 2), dtype=np.float64)
                in_coords = True
            elif tok[0] == "EOF":
                break
            elif in_coords and tok[0].lstrip("+-").isdigit():
                i = int(tok[0]) - 1
                if 0 <= i < n and len(tok) >= 3:
                    xy[i, 0] = float(tok[1])
                    xy[i, 1] = float(tok[2])
    if xy is None:
        raise ValueError("%s: no NODE_COORD_SECTION" % path)
    return xy, wt
