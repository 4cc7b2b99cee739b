"""GPU, self-arming pin of the neural path on the ONLY result the reference publishes: the title of `pics/demo2.png`,
"Overlap: [0.8919191]  Yaw: [0]", produced by demo2 with the authors' trained `model_geo.weight` (README.md:120 gives a download
link; the file is git-ignored upstream, .gitignore:9, and is not in this tree).

    OVERLAPNET_MODEL_GEO=/path/to/model_geo.weight python -m pytest tests/test_demo2_published.py -m gpu -q

Without that variable (or without a GPU) the test is collected and SKIPPED.  With it, it follows demo1 + demo2 step by step:
  * demo1 (demo1_gen_data.py -> gen_depth_data.py:10-48, gen_normal_data.py:10-46): the two shipped scans data/scans/00000{0,1}.bin
    (embedded in tests/golden/kitti_preprocess.npz) -> depth/ and normal/ .npy files of sequence "preprocess_data_demo";
  * demo2 (demo2_infer.py:52-69): config/network.yml as shipped (depth + normals, 360OutputkLegs, DeltaLayerConv1NetworkHead,
    CorrelationHead), `infer_seqs = Demo2.infer_seqs`, and -- note the swap at demo2_infer.py:69 --
    `Infer(config).infer_one(scan2_path, scan1_path)`; inside, infer.py:138-152 makes scan1 the LEFT input of the head.
The assertion is north_star's tolerance: |overlap - 0.8919191| <= 1e-4 and yaw == 0, in the default (f16x3) and the fp32 mode."""
import os

import numpy as np
import pytest
import torch

WEIGHTS = os.environ.get("OVERLAPNET_MODEL_GEO", "")

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X")]
needs_model_geo = pytest.mark.skipif(not WEIGHTS, reason="set OVERLAPNET_MODEL_GEO=/path/to/model_geo.weight (README.md:120 of the "
                                                         "reference) to pin the neural path on the published demo2 result")

PUBLISHED_OVERLAP = 0.8919191     # pics/demo2.png
PUBLISHED_YAW = 0


def network_yml(data_root, weightfile):
    """config/network.yml of the reference, as a dict (the keys `Infer` reads: infer.py:31-93,117)."""
    return {"pretrained_weightsfilename": weightfile,
            "use_depth": True, "use_normals": True, "use_class_probabilities": False, "use_class_probabilities_pca": False,
            "use_intensity": False, "data_root_folder": str(data_root), "infer_seqs": "preprocess_data_demo", "batch_size": 16,
            "model": {"modelType": "SiameseNetworkTemplate", "legsType": "360OutputkLegs",
                      "overlap_head": "DeltaLayerConv1NetworkHead", "orientation_head": "CorrelationHead",
                      "inputShape": [64, 900], "leg_output_width": 360, "strides_layer1": [2, 2],
                      "additional_unsymmetric_layer3a": True}}


def run_demo2(tmp_path, fx, weightfile, precision=None):
    """demo1 on the two shipped scans, then demo2; returns (Infer, overlap, yaw)."""
    from overlapnet_amd import preprocess as P
    from overlapnet_amd.infer import Infer
    scans = tmp_path / "data" / "scans"
    os.makedirs(scans)
    fx["points_0"].astype(np.float32).tofile(scans / "000000.bin")
    fx["points_1"].astype(np.float32).tofile(scans / "000001.bin")
    dst = tmp_path / "data" / "preprocess_data_demo"
    os.makedirs(dst)
    P.gen_depth_data(str(scans), str(dst))        # demo1
    P.gen_normal_data(str(scans), str(dst))
    cfg = network_yml(tmp_path / "data", weightfile)
    if precision:
        cfg["precision"] = precision              # extension key of this framework: every contraction on the fp32 matrix cores
    infer = Infer(cfg)
    scan1_path, scan2_path = str(scans / "000000.bin"), str(scans / "000001.bin")
    overlap, yaw = infer.infer_one(scan2_path, scan1_path)        # demo2_infer.py:69 passes (scan2, scan1)
    assert list(infer.filenames) == ["000000", "000001"]          # infer.py:148: [name2, name1] -> what demo2 plots as scan1, scan2
    assert overlap.shape == (1,) and yaw.shape == (1,)
    return infer, overlap, yaw


@needs_model_geo
@pytest.mark.parametrize("precision", [None, "f32"])
def test_demo2_reproduces_the_published_overlap_and_yaw(tmp_path, fixture_npz, precision):
    assert os.path.isfile(WEIGHTS), "OVERLAPNET_MODEL_GEO=%s is not a file" % WEIGHTS
    _, overlap, yaw = run_demo2(tmp_path, fixture_npz, WEIGHTS, precision)
    print("demo2: overlap %r yaw %r (published: [%s] [%d])" % (overlap, yaw, PUBLISHED_OVERLAP, PUBLISHED_YAW))
    assert abs(float(overlap[0]) - PUBLISHED_OVERLAP) <= 1e-4
    assert int(yaw[0]) == PUBLISHED_YAW


def test_demo2_flow_on_a_keras_layout_file_with_seeded_weights(tmp_path, fixture_npz):
    """The same demo1 + demo2 flow, always run: a full-size Keras-layout HDF5 file written by real h5py with SEEDED weights
    (tests/golden/make_keras_full_golden.py) stands in for model_geo.weight; expected values come from the fp64 oracle with the
    roles of infer.py:138-152 (scan1 = file 000000 -> head left, scan2 = 000001 -> head right)."""
    from oracle import overlapnet_oracle as O
    from tools import synthetic as S
    from overlapnet_amd import weights as W
    wfile = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keras_layout_full_c4.weight")
    _, overlap, yaw = run_demo2(tmp_path, fixture_npz, wfile)
    w = W.load_weights_file(wfile)
    imgs = np.stack([S.stack(fixture_npz["range_%d" % i], fixture_npz["normal_%d" % i], None, (True, True, False)) for i in range(2)])
    fv = O.leg_forward(imgs, w, S.REFERENCE_MODEL_CFG, np.float64)
    o_ov, o_yaw, _, _ = O.heads_forward(fv[[0]], fv[[1]], w)
    assert abs(float(overlap[0]) - o_ov[0]) <= 1e-4 and int(yaw[0]) == int(o_yaw[0])
