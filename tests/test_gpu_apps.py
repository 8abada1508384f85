"""The command-line callers of the path: generate_data and
super_resolution (host/apps) run end to end on a GPU box on a small synthetic
cube in ENVI format, and agree with the library driven from Python."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _write_envi(path, cube):
    cube = np.asarray(cube, dtype="<f4")
    cube.tofile(path)
    C, H, W = cube.shape
    with open(path + ".config", "w") as f:
        f.write("file %s\ninterleave bsq\ndata_type float\nbig_endian false\nheader_offset 0\n" % path)
        f.write("num_data_rows %d\nnum_data_cols %d\nnum_data_bands %d\n" % (H, W, C))
        f.write("start_row 0\nend_row %d\nstart_col 0\nend_col %d\nstart_band 0\nend_band %d\n" % (H, W, C))
    return path + ".config"


def _read_envi(path, shape):
    return np.fromfile(path, dtype="<f4").reshape(shape).astype(np.float64)


def _ground_truth(C, H, W):
    v, u = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
    base = 0.5 + 0.25 * np.sin(2 * np.pi * 2 * u) * np.cos(2 * np.pi * 3 * v) + 0.2 * ((u - .5) ** 2 + (v - .5) ** 2 < .08)
    return np.stack([np.clip(base * (0.7 + 0.3 * c / max(1, C - 1)), 0, 1) for c in range(C)])


def test_generate_then_super_resolve(tmp_path):
    import __graft_entry__ as ge
    ge.build_lib()
    gen, sr = ge.build_apps()
    C, H, W, s, K = 2, 48, 64, 2, 4
    gt = _ground_truth(C, H, W)
    gt_cfg = _write_envi(str(tmp_path / "gt"), gt)
    motion = tmp_path / "motion.txt"
    motion.write_text("0 0\n1 1\n0 1\n1 0\n")
    lr_dir = tmp_path / "lr"
    lr_dir.mkdir()
    out = subprocess.run([gen, "--input_image=" + gt_cfg, "--output_image_dir=" + str(lr_dir),
                          "--motion_sequence_path=" + str(motion), "--blur_radius=3", "--blur_sigma=1.0",
                          "--downsampling_scale=%d" % s, "--number_of_frames=%d" % K],
                         capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0
    frames = np.stack([_read_envi(str(lr_dir / ("low_res_%d" % i)), (C, H // s, W // s)) for i in range(K)])

    # the frames on disk are the image model applied by the library (float32 on disk)
    import srmap
    ctx = srmap.Context(0)
    shifts = [[0, 0], [1, 1], [0, 1], [1, 0]]
    prob = srmap.Problem(ctx, W, H, C, K, s, shifts, 3, 1.0, srmap.F64)
    gt32 = gt.astype(np.float32).astype(np.float64)   # what the tool read back from the ENVI file
    for k in range(K):
        assert np.allclose(frames[k], prob.apply(gt32, k), atol=2e-7)

    result_path = str(tmp_path / "result")
    out = subprocess.run([sr, "--data_path=" + str(lr_dir), "--ground_truth_image=" + gt_cfg,
                          "--upsampling_scale=%d" % s, "--blur_radius=3", "--blur_sigma=1.0",
                          "--motion_sequence_path=" + str(motion), "--regularizer=btv", "--btv_scale_range=2",
                          "--regularization_parameter=0.001", "--optimization_iterations=5",
                          "--solver_iterations=30", "--evaluators=psnr", "--result_path=" + result_path,
                          "--save_initial_estimate=" + str(tmp_path / "x0.f64")],
                         capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0
    lines = {l.split(":")[0].strip(): float(l.split(":")[1]) for l in out.stdout.splitlines() if l.startswith("PSNR")}
    assert lines["PSNR score on result"] > lines["PSNR score on upsampled"] + 1.0
    result = _read_envi(result_path, (C, H, W))
    mse = np.mean((result - gt32) ** 2)
    assert abs(-10 * np.log10(mse) - lines["PSNR score on result"]) < 1e-3

    # same solve through the Python binding of the same C ABI: same image up to the float32 file format
    prob.set_observations(frames)
    prob.add_regularizer(srmap.REG_BTV, 0.001, 2, 0.5)
    o = srmap.default_irls_options()
    o.max_num_irls_iterations, o.max_num_solver_iterations = 5, 30
    import bench
    x0 = np.stack([bench.bilinear_upsample(frames[0, c:c + 1], s)[0] for c in range(C)])
    x, _ = prob.solve(x0, o)
    assert np.max(np.abs(x - result)) < 5e-3

    # and through the CPU oracle (the reference's algorithm, ALGLIB-driven when oracle/_ref is built), started from the
    # IDENTICAL x0 the tool used (--save_initial_estimate: raw float64): the image the reference binary would have
    # written, up to the float32 file format of the result -- a wrong lambda or weight would show at 1e-3
    import oracle as orc
    x0_cli = np.fromfile(str(tmp_path / "x0.f64"), dtype=np.float64).reshape(C, H, W)
    print("tool's bilinear start vs the harness's: max |diff| %.2e" % np.max(np.abs(x0_cli - x0)))
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
    ref = orc.Problem(model, frames)
    ref.add_regularizer(orc.REG_BTV, 0.001, 2, 0.5)
    oo = orc.default_irls_options()
    oo.max_num_irls_iterations, oo.max_num_solver_iterations = 5, 30
    x_ref, rep_ref = ref.solve(x0_cli, oo, use_alglib=orc.have_ref())
    psnr_cli, psnr_ref = -10 * np.log10(np.mean((result - gt32) ** 2)), orc.psnr(gt32, x_ref)
    print("CLI result vs oracle solve from the same x0: max |diff| %.2e, PSNR %.4f / %.4f dB" % (
        np.max(np.abs(result - x_ref)), psnr_cli, psnr_ref))
    assert np.max(np.abs(result - x_ref)) < 2e-6
    assert abs(psnr_cli - psnr_ref) < 0.001


def test_pgm_round_trip_and_usage(tmp_path):
    import __graft_entry__ as ge
    ge.build_lib()
    gen, sr = ge.build_apps()
    img = (np.arange(20 * 30).reshape(20, 30) % 251).astype(np.uint8)
    p = tmp_path / "a.pgm"
    with open(p, "wb") as f:
        f.write(b"P5\n# comment\n30 20\n255\n" + img.tobytes())
    q = tmp_path / "b.pgm"
    out = subprocess.run([gen, "--input_image=" + str(p), "--save_as=" + str(q)], capture_output=True, text=True)
    assert out.returncode == 0
    data = open(q, "rb").read()
    assert data.endswith(img.tobytes())
    bad = subprocess.run([sr, "--no_such_flag=1", "--data_path=x"], capture_output=True, text=True)
    assert bad.returncode == 2 and "unknown flag" in bad.stderr


def test_super_resolve_in_pca_space(tmp_path):
    """--solve_in_pca_space: project the frames, solve, reconstruct (super_resolution.cpp:344-366, 398-400).
    The data term is invariant under the orthogonal change of basis as long as the image model maps constant
    images to constant images (the PCA mean is subtracted before the model's zero-filled warp and zero-padded
    blur see the data: with shifts or blur the reference's flow is inconsistent at the image border, and so is
    this one).  Without shifts, blur and regulariser the PCA-space solve must therefore reproduce the plain one."""
    import __graft_entry__ as ge
    ge.build_lib()
    gen, sr = ge.build_apps()
    C, H, W, s, K = 6, 32, 48, 2, 2
    rng = np.random.default_rng(5)
    base = _ground_truth(C, H, W)
    gt = np.clip(base + 0.05 * rng.standard_normal(base.shape) + 0.1 * rng.random((C, 1, 1)), 0, 1)  # full-rank cube
    gt_cfg = _write_envi(str(tmp_path / "gt"), gt)
    motion = tmp_path / "motion.txt"
    motion.write_text("0 0\n0 0\n")
    common = ["--data_path=" + gt_cfg, "--generate_lr_images", "--number_of_frames=%d" % K,
              "--upsampling_scale=%d" % s, "--blur_radius=0", "--blur_sigma=0",
              "--motion_sequence_path=" + str(motion), "--regularization_parameter=0",
              "--optimization_iterations=2", "--solver_iterations=20", "--evaluators=psnr"]
    results = {}
    for name, extra in (("full", []), ("pca", ["--solve_in_pca_space"]), ("pca4", ["--solve_in_pca_space", "--num_pca_components=4"])):
        path = str(tmp_path / ("res_" + name))
        out = subprocess.run([sr] + common + extra + ["--result_path=" + path], capture_output=True, text=True, timeout=600)
        print(out.stdout, out.stderr)
        assert out.returncode == 0
        results[name] = _read_envi(path, (C, H, W))
        if name == "pca4":
            assert "PCA space with 4 PCA components" in out.stdout
    assert np.max(np.abs(results["pca"] - results["full"])) < 1e-4
    # dropping two of six components loses little on this cube, but something
    err4 = np.sqrt(np.mean((results["pca4"] - results["full"]) ** 2))
    assert 0 < err4 < 0.1


def test_interpolate_color_path(tmp_path):
    """--interpolate_color: the frames go to YCrCb, only the luminance is super-resolved, Cr / Cb are bilinearly
    interpolated and the result returns to BGR (super_resolution.cpp:330-342, 392-395).  PPM in, PPM out."""
    import __graft_entry__ as ge
    ge.build_lib()
    gen, sr = ge.build_apps()
    H, W, s, K = 48, 64, 2, 4
    gt = _ground_truth(3, H, W)
    gt[1] = np.clip(gt[1] * 0.8 + 0.1, 0, 1)
    gt[2] = np.clip(1.0 - gt[2], 0, 1)
    rgb = np.round(np.transpose(gt, (1, 2, 0)) * 255).astype(np.uint8)
    p = tmp_path / "gt.ppm"
    with open(p, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (W, H) + rgb.tobytes())
    motion = tmp_path / "motion.txt"
    motion.write_text("0 0\n1 1\n0 1\n1 0\n")
    common = [sr, "--data_path=" + str(p), "--generate_lr_images", "--number_of_frames=%d" % K, "--upsampling_scale=%d" % s,
              "--blur_radius=3", "--blur_sigma=1.0", "--motion_sequence_path=" + str(motion), "--regularizer=tv",
              "--regularization_parameter=0.001", "--optimization_iterations=4", "--solver_iterations=30", "--evaluators=psnr"]
    scores = {}
    for name, extra in (("all", []), ("luma", ["--interpolate_color"])):
        q = tmp_path / (name + ".ppm")
        out = subprocess.run(common + extra + ["--result_path=" + str(q)], capture_output=True, text=True, timeout=600)
        print(out.stdout, out.stderr)
        assert out.returncode == 0
        scores[name] = {l.split(":")[0].strip(): float(l.split(":")[1]) for l in out.stdout.splitlines() if l.startswith("PSNR")}
        data = open(q, "rb").read()
        assert data.startswith(b"P6") and len(data) >= H * W * 3
    assert "only the luminance channel" in out.stdout
    # solving all three channels is the upper bound; luminance-only must still beat plain bilinear upsampling
    assert scores["all"]["PSNR score on result"] > scores["luma"]["PSNR score on result"] > scores["luma"]["PSNR score on upsampled"]


def test_generate_without_motion_file(tmp_path):
    """generate_data with the reference CLI defaults (several frames, NO motion file): blur and downsampling ignore
    the frame index (blur_module.cpp:25-28, downsampling_module.cpp:19-27), so every frame is the same image."""
    import __graft_entry__ as ge
    ge.build_lib()
    gen, _ = ge.build_apps()
    C, H, W, s, K = 1, 24, 32, 2, 4
    gt = _ground_truth(C, H, W)
    gt_cfg = _write_envi(str(tmp_path / "gt"), gt)
    lr_dir = tmp_path / "lr"
    lr_dir.mkdir()
    out = subprocess.run([gen, "--input_image=" + gt_cfg, "--output_image_dir=" + str(lr_dir),
                          "--blur_radius=3", "--blur_sigma=1.0", "--downsampling_scale=%d" % s,
                          "--number_of_frames=%d" % K], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0
    frames = np.stack([_read_envi(str(lr_dir / ("low_res_%d" % i)), (C, H // s, W // s)) for i in range(K)])
    import srmap
    prob = srmap.Problem(srmap.Context(0), W, H, C, 1, s, None, 3, 1.0, srmap.F64)
    ref = prob.apply(gt.astype(np.float32).astype(np.float64), 0)
    for k in range(K):
        assert np.allclose(frames[k], ref, atol=2e-7)


def test_noise_module_and_ssim_flag(tmp_path):
    """--noise_sigma runs the AdditiveNoiseModule at the end of the generating chain (sigma / 255 per pixel, seeded);
    --evaluators=psnr,ssim prints both metrics (super_resolution.cpp:405-430)."""
    import __graft_entry__ as ge
    ge.build_lib()
    gen, sr = ge.build_apps()
    C, H, W, s, K = 1, 64, 64, 2, 4
    gt = _ground_truth(C, H, W)
    gt_cfg = _write_envi(str(tmp_path / "gt"), gt)
    motion = tmp_path / "motion.txt"
    motion.write_text("0 0\n1 1\n0 1\n1 0\n")
    frames = {}
    for tag, sigma in (("clean", 0), ("noisy", 8)):
        d = tmp_path / tag
        d.mkdir()
        out = subprocess.run([gen, "--input_image=" + gt_cfg, "--output_image_dir=" + str(d),
                              "--motion_sequence_path=" + str(motion), "--downsampling_scale=%d" % s,
                              "--number_of_frames=%d" % K, "--noise_sigma=%g" % sigma, "--noise_seed=3"],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        frames[tag] = np.stack([_read_envi(str(d / ("low_res_%d" % i)), (C, H // s, W // s)) for i in range(K)])
    diff = frames["noisy"] - frames["clean"]
    assert abs(diff.std() - 8 / 255) < 0.1 * 8 / 255 and abs(diff.mean()) < 4 * (8 / 255) / np.sqrt(diff.size)
    out = subprocess.run([sr, "--data_path=" + str(tmp_path / "noisy"), "--ground_truth_image=" + gt_cfg,
                          "--upsampling_scale=%d" % s, "--blur_radius=0", "--motion_sequence_path=" + str(motion),
                          "--regularizer=tv", "--regularization_parameter=0.002", "--optimization_iterations=3",
                          "--solver_iterations=20", "--evaluators=psnr,ssim"],
                         capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0
    vals = {l.split(":")[0].strip(): float(l.split(":")[1]) for l in out.stdout.splitlines() if "score on" in l}
    assert set(vals) == {"PSNR score on upsampled", "PSNR score on result", "SSIM score on upsampled", "SSIM score on result"}
    assert 0.0 < vals["SSIM score on result"] <= 1.0 and vals["SSIM score on result"] >= vals["SSIM score on upsampled"] - 0.02
