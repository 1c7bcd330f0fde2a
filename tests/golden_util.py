"""Load the committed golden fixtures (tests/golden/*.npz) into the structures the trackers / EF classes take."""
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_tracker():
    return np.load(os.path.join(HERE, "tracker_small.npz"))


def setup_tracker(T, g):
    fx, fy, cx, cy = g["calib"]
    T.makeK(float(fx), float(fy), float(cx), float(cy))
    for l in range(3):
        T.set_ref(l, g["ref%d_u" % l], g["ref%d_v" % l], g["ref%d_idepth" % l], g["ref%d_color" % l])
    T.set_ref_frame(1.0, float(g["ref_aff"][0]), float(g["ref_aff"][1]))
    T.set_new_image(g["image"], 1.0)
    return T


class _W:
    pass


def load_window():
    from sdv_loam_amd import synthetic as syn
    g = np.load(os.path.join(HERE, "backend_small.npz"))
    W = _W()
    for k in g.files:
        setattr(W, k, g[k])
    for k in ("w", "h", "nF", "nP", "nR"):
        setattr(W, k, int(g[k]))
    W.images = [g["images"][i] for i in range(W.nF)]
    W.pyr0 = [syn.pyramid_numpy(img, 1)[0] for img in W.images]
    return W, g


def load_struct_pose():
    g = np.load(os.path.join(HERE, "struct_pose_small.npz"))
    fx, fy, cx, cy = (float(x) for x in g["calib"])
    args = (g["u"], g["v"], g["idepth"], g["host_idx"], g["host_poses7"], g["obs"])
    return g, dict(fx=fx, fy=fy, cx=cx, cy=cy), args


def load_reproject():
    """Returns (g, setup) where setup(T) registers calib, key-frames and the new frame on an Oracle/GPU reprojector.  Only the
    intensities are stored; the {I,dx,dy} images are rebuilt with the (bit-exact, tested) numpy pyramid mirror."""
    from sdv_loam_amd import synthetic as syn
    g = np.load(os.path.join(HERE, "reproject_small.npz"))
    fx, fy, cx, cy = (float(x) for x in g["calib"])
    levels = int(g["levels"])

    def setup(T):
        T.set_calib(fx, fy, cx, cy)
        for k in range(len(g["frame_poses7"])):
            T.set_frame(k, g["frame_poses7"][k], syn.pyramid_numpy(g["frame_I"][k], 1)[0], 1.0, 0.01 * k, 0.3 * k)
        T.set_cur(g["cur_pose7"], syn.pyramid_numpy(g["cur_I"], levels), 1.0, 0.02, 1.0)
        return T
    return g, setup


def load_marginalize():
    return np.load(os.path.join(HERE, "marginalize_small.npz"))
