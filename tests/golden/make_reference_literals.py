"""Transcribes the known-answer literals of the reference's own gtest files
into tests/golden/reference_literals.json (data only: inputs and expected
outputs, each tagged with the reference file:line it comes from), and decodes
the reference's 28x28 test icon (test_data/fb.png) into fb_gray.json with the
same conversion cv::imread(..., CV_LOAD_IMAGE_GRAYSCALE) + ImageData's 1/255
normalisation performs.  Run once in the build container:

    python tests/golden/make_reference_literals.py
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

lit = {}

# test/test_image_model.cpp:22-27 (kSmallTestImage), :173-193, :197-225
small = [[1, 2, 3, 4, 5, 6], [7, 8, 9, 0, 1, 2], [9, 7, 5, 4, 2, 1], [2, 4, 6, 8, 0, 1]]
lit["small_test_image"] = {"src": "test/test_image_model.cpp:22-27", "data": small}
lit["downsample_scale2"] = {"src": "test/test_image_model.cpp:188-193",
                            "expected": [[1, 3, 5], [9, 5, 2]]}
lit["downsample_transpose_scale2"] = {
    "src": "test/test_image_model.cpp:197-225",
    "expected": [[1, 0, 2, 0, 3, 0, 4, 0, 5, 0, 6, 0], [0] * 12,
                 [7, 0, 8, 0, 9, 0, 0, 0, 1, 0, 2, 0], [0] * 12,
                 [9, 0, 7, 0, 5, 0, 4, 0, 2, 0, 1, 0], [0] * 12,
                 [2, 0, 4, 0, 6, 0, 8, 0, 0, 0, 1, 0], [0] * 12]}
# test/test_image_model.cpp:350-408
lit["blur_3_0.849321"] = {
    "src": "test/test_image_model.cpp:350-408", "ksize": 3, "sigma": 0.849321, "tol": 0.001,
    "expected": [[1.875, 3.0, 3.125, 2.625, 2.75, 2.4375],
                 [4.5625, 6.25, 5.3125, 3.1875, 2.3125, 1.9375],
                 [5.0, 6.5, 5.75, 3.875, 1.9375, 0.9375],
                 [2.5625, 3.75, 4.3125, 3.6875, 1.6875, 0.5]]}
# test/test_image_model.cpp:289-347: MotionShift -> operator matrix (3x3 image)
lit["motion_matrices_3x3"] = {
    "src": "test/test_image_model.cpp:289-347",
    "shifts": [[0, 0], [1, 1], [-1, 0]],
    # (output index <- input index) pairs with a 1 in the matrix
    "ones": [[[i, i] for i in range(9)],
             [[4, 0], [5, 1], [7, 3], [8, 4]],
             [[0, 1], [1, 2], [3, 4], [4, 5], [6, 7], [7, 8]]]}
# test/test_image_data.cpp:311-403
img44 = [[0.1, 0.2, 0.3, 0.4], [0.5, 0.6, 0.7, 0.8], [0.9, 1.0, 0.0, 0.2], [0.4, 0.6, 0.8, 1.0]]
lit["resize"] = {
    "src": "test/test_image_data.cpp:311-403", "image": img44,
    "nearest_down_2x2": [[0.1, 0.3], [0.9, 0.0]],
    "nearest_up_8x8": np.repeat(np.repeat(np.array(img44), 2, axis=0), 2, axis=1).tolist(),
    "additive_up_8x8": np.kron(np.array(img44), np.array([[1, 0], [0, 0]])).tolist(),
    "additive_down_2x2": [[0.1 + 0.2 + 0.5 + 0.6, 0.3 + 0.4 + 0.7 + 0.8],
                          [0.9 + 1.0 + 0.4 + 0.6, 0.0 + 0.2 + 0.8 + 1.0]]}
# test/test_tv_regularizer.cpp:20-45, :76-145
tv_img = [0, 0, 1, 0, 1, 3, -3, -1, 0]
lit["tv"] = {"src": "test/test_tv_regularizer.cpp:20-73", "size": [3, 3], "image": tv_img,
             "expected": [0, 2, 2, 4, 4, 3, 2, 1, 0],
             "fd_step": 1e-6, "fd_tol": 1e-4, "fd_src": "test/test_tv_regularizer.cpp:150-198"}
lit["tv3d"] = {"src": "test/test_tv_regularizer.cpp:76-145", "size": [3, 3],
               "image": [0, 0, 1, 0, 1, 3, -3, -1, 0,
                         0, 0, 0, 0, 0, 0, 0, 0, 0,
                         0, -1, 2, -3, 4, 5, 6, 7, -8],
               "expected": [0, 2, 3, 4, 5, 6, 5, 2, 0,
                            0, 1, 2, 3, 4, 5, 6, 7, 8,
                            4, 8, 3, 16, 4, 13, 1, 15, 0]}
# test/test_btv_regularizer.cpp:12-95
btv_img = [0, 0, 1, 2, 1, 0, 1, 3, 2, 3, 5, 4, 3, -2, 1, 4, 6, 9, 3, 0, -3, -1, 0, 6, 0]
lit["btv"] = {"src": "test/test_btv_regularizer.cpp:12-72", "size": [5, 5], "image": btv_img,
              "case_range2_decay0.5": {"index0": 2.8125, "index24": 0.0},
              "case_range1_decay0.25_two_channels": {"index7": 0.5625, "index32": 0.5625,
                                                     "index24": 0.0, "index49": 0.0}}
# test/test_evaluation.cpp:12-96
gt = [[0.0, 0.1, 0.2, 0.3], [0.7, 0.6, 0.5, 0.4], [0.8, 0.9, 1.0, 0.5], [0.4, 0.6, 0.0, 1.0]]
lit["psnr"] = {"src": "test/test_evaluation.cpp:12-47", "ground_truth": gt,
               "modified": {"6": 0.25, "15": 0.5}, "expected": 17.09269960975831,
               "image3": [[0.2, 0.9, 1.0, 0.0], [0.7, 0.0, 0.8, 0.3], [0.1, 0.0, 0.2, 1.0], [0.0, 0.5, 0.5, 0.3]]}
# test/test_map_solver.cpp:79-199
lit["map_solver_small_data"] = {
    "src": "test/test_map_solver.cpp:79-199", "scale": 2,
    "lr_values": [0.4, 0.2, 0.0, 1.0], "lr_size": [2, 2],
    "shifts": [[0, 0], [-1, 0], [0, -1], [-1, -1]],
    "expected": [[0.4, 0.2, 0.4, 0.2], [0.0, 1.0, 0.0, 1.0], [0.4, 0.2, 0.4, 0.2], [0.0, 1.0, 0.0, 1.0]],
    "tol": 0.001, "channels_multi": 10}
# test/test_map_solver.cpp:205-308 (RealIconDataTest) parameters
lit["map_solver_icon"] = {"src": "test/test_map_solver.cpp:205-308", "scale": 2,
                          "shifts": [[0, 0], [1, 0], [0, 1], [1, 1]],
                          "roi": [1, 1, 26, 26], "tol": 0.001}
# test_data/test_motion_sequence_{4,9}.txt
lit["motion_sequence_4"] = {"src": "test_data/test_motion_sequence_4.txt",
                            "shifts": [[0, 0], [1, 1], [0, 1], [1, 0]]}
lit["motion_sequence_9"] = {"src": "test_data/test_motion_sequence_9.txt",
                            "shifts": [[i, j] for i in range(3) for j in range(3)]}

with open(os.path.join(HERE, "reference_literals.json"), "w") as f:
    json.dump(lit, f, indent=1)

ref_png = "/root/reference/test_data/fb.png"
if os.path.exists(ref_png):
    from PIL import Image
    im = Image.open(ref_png)
    rgb = np.asarray(im.convert("RGB"), dtype=np.int64)
    # cv::imread(GRAYSCALE): BT.601 fixed point, Y = (R*4899 + G*9617 + B*1868 + 8192) >> 14
    gray = (rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14
    with open(os.path.join(HERE, "fb_gray.json"), "w") as f:
        json.dump({"src": "test_data/fb.png (decoded, 8-bit gray; divide by 255)",
                   "mode": im.mode, "size": list(gray.shape), "data": gray.tolist()}, f)
    print("fb.png", im.mode, gray.shape, gray.min(), gray.max())
