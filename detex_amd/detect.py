"""Host side of the detection scan: the reference's trigger-picking loop
(detex/detect.py:390-445, 526-557) run on the sparse list of above-threshold
samples the GPU returns instead of on the full detection-statistic vector."""
import numpy as np


def pick_triggers_sparse(idx, val, P, threshold, sr, maxds, buff=20, maxcount=4000):
    """Exact restatement of ``_evalTrigCon`` + ``_CreateCoeffArray`` +
    ``_downPlayArrayAroundMax`` (trigCon == 0) on sparse candidates.

    idx / val : window indices (ascending) and DS values of every sample with
    DS >= threshold in one data block of P windows; maxds : DS.max() of the block.
    Returns [(trigIndex, DS), ...] in the order the reference finds them.

    Only samples >= threshold can ever be the arg-max while the loop runs
    (detect.py:410), and zeroing a range only removes candidates, so the sparse
    loop visits exactly the indices the dense loop does; ``argmax`` picks the
    first occurrence of the maximum in both (idx is ascending)."""
    out = []
    if not (maxds > threshold):                      # detect.py:534, strict '>'
        return out
    idx = np.asarray(idx, dtype=np.int64)
    val = np.array(val, dtype=np.float64)
    alive = np.ones(len(idx), dtype=bool)
    w = buff * sr
    count = 0
    while alive.any():
        cur = np.where(alive, val, -np.inf)
        j = int(np.argmax(cur))                      # first occurrence of the max
        if not (cur[j] >= threshold):                # detect.py:410
            break
        index = int(idx[j])
        out.append((index, float(val[j])))
        # detect.py:551-556
        if index < w + 1:
            lo, hi = 0, int(index + w)
        elif index > P - w:
            lo, hi = int(index - w), P
        else:
            lo, hi = int(index - w), int(w + index)
        alive &= ~((idx >= lo) & (idx < hi))
        if count > maxcount:                         # detect.py:433-436
            raise RuntimeError('over 4000 events found in single data block')
        count += 1
    return out
