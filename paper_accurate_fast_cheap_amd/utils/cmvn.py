"""Global CMVN statistics loaders (reference: wenet/utils/cmvn.py:20-93): JSON {mean_stat,var_stat,frame_num}
or Kaldi text `[ sums... count \\n sumsq... 0 ]` -> (mean, istd) with the 1e-20 variance floor."""
import json
import math

import numpy as np


def _finish(sums, sumsq, count):
    mean = np.asarray(sums, dtype=np.float64) / count
    var = np.asarray(sumsq, dtype=np.float64) / count - mean * mean
    var = np.maximum(var, 1.0e-20)
    return mean, 1.0 / np.sqrt(var)


def load_cmvn(cmvn_file: str, is_json: bool):
    if is_json:
        with open(cmvn_file) as f:
            st = json.load(f)
        return _finish(st["mean_stat"], st["var_stat"], st["frame_num"])
    with open(cmvn_file, "r") as f:
        if f.read(2) == "\0B":
            raise ValueError("binary Kaldi cmvn is not supported; recompute with --binary=false")
        f.seek(0)
        arr = f.read().split()
    assert arr[0] == "[" and arr[-2] == "0" and arr[-1] == "]"
    dim = (len(arr) - 4) // 2
    sums = [float(a) for a in arr[1:dim + 1]]
    count = float(arr[dim + 1])
    sumsq = [float(a) for a in arr[dim + 2:2 * dim + 2]]
    return _finish(sums, sumsq, count)
