"""Golden vectors of ``get_rot_mat`` (core/gdrn_modeling/models/model_utils.py:347-359) for every ROT_TYPE family, from the
reference's own functions imported from source (authoring container only): quat2mat_torch (core/utils/pose_utils.py),
quaternion_lf.qexp (core/utils/quaternion_lf.py), lie_algebra.lie_vec_to_rot (core/utils/lie_algebra.py),
rot6d_to_mat_batch (core/utils/rot_reps.py).  Inputs include the small-angle branches (|v| -> 0)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import _refimport  # noqa: E402

_refimport.install()
from core.gdrn_modeling.models.model_utils import get_rot_mat  # noqa: E402

from gdrnpp_bop2022_amd import synthetic as S  # noqa: E402


def main():
    n = 64
    rec = {}
    for name, dim, rot_type in (("quat", 4, "allo_quat"), ("log_quat", 3, "ego_log_quat"), ("lie_vec", 3, "allo_lie_vec"),
                                ("rot6d", 6, "ego_rot6d")):
        x = S.seeded_uniform("rot_golden." + name, (n, dim), 7) * np.float32(2.5)
        if dim == 3:
            x[:4] *= np.float32(1e-4)       # theta^2 <= 1e-6: first-order branch of lie_vec_to_rot, sin(theta)/theta -> 1 of qexp
            x[4] = 0.0
            x[5] = np.float32([3.1, 0.2, -0.1])   # near pi
        rec[name + "_in"] = x
        rec[name + "_R"] = get_rot_mat(torch.from_numpy(x.copy()), rot_type).numpy()
        print(name, rec[name + "_R"].shape, float(np.abs(rec[name + "_R"]).max()))
    np.savez_compressed(os.path.join(HERE, "rot_golden.npz"), **rec)
    print("wrote rot_golden.npz")


if __name__ == "__main__":
    main()
