"""SURVEY.md section 8 row f1, training side: the GPU augmentation stages (csrc/augment.hip) against the numpy restatement
of the reference's TRAIN_PIPELINE (oracle/transforms.py::train_pipeline; main_both.py:407-429)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STEP = 1.0 / 255 / 0.224


def _pipe(h, w, n, px):
    from garbage_classification_rca_amd.preprocess import GpuImagePipeline
    return GpuImagePipeline(h, w, max_batch=n, max_pixels=px)


def _smooth(rng, h, w):
    """Low-frequency image + noise: has edges and flat regions like a photo (pure noise would hide interpolation errors in
    the rounding noise; a constant image would hide everything)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 127 + 80 * np.sin(xx / (5 + 20 * rng.random())) * np.cos(yy / (7 + 20 * rng.random()))
    img = base[..., None] + rng.normal(0, 25, (h, w, 3))
    img[h // 3: h // 2, w // 4: w // 2] = rng.integers(0, 256, 3)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("name,param", [
    ("blur3", dict(blur_k=3)), ("blur5", dict(blur_k=5)), ("blur7", dict(blur_k=7)),
    ("flips", dict(flip_v=True, flip_h=True)), ("blur_flip_v", dict(blur_k=5, flip_v=True)),
    ("bc_up", dict(bc=(1.17, 0.12))), ("bc_down", dict(bc=(0.83, -0.19))),
    ("sharpen", dict(sharpen=(0.37, 0.81))),
    ("persp", dict(persp=np.array([[0.07, 0.03], [0.11, 0.06], [0.02, 0.09], [0.05, 0.13]]))),
    ("zoom_out", dict(scale=0.62)), ("zoom_in", dict(scale=1.41)),
    ("all_post", dict(blur_k=3, flip_h=True, bc=(1.1, -0.05), sharpen=(0.25, 0.6),
                      persp=np.array([[0.04, 0.08], [0.09, 0.02], [0.06, 0.06], [0.03, 0.1]]), scale=0.8)),
])
def test_single_stages_are_bit_exact_on_network_sized_inputs(name, param):
    """Inputs already have the network's size, so pad + resize is the identity and every later stage sees exactly the
    oracle's uint8 image: the stages' arithmetic (1/32-pixel coordinates, reflect-101 filters, rounding rules) must then
    agree with the oracle bit for bit."""
    from oracle import transforms as T
    rng = np.random.default_rng(abs(hash(name)) % 1000)
    H = W = 96
    imgs = [_smooth(rng, H, W) for _ in range(3)]
    out = _pipe(H, W, 4, H * W)(imgs, aug=[param] * 3).cpu().numpy()
    for b, img in enumerate(imgs):
        ref = T.train_pipeline(img, H, W, param)
        d = np.abs(out[b] - ref)
        if "persp" in param:        # the product's closed-form homography and the oracle's 8x8 solve agree to ~1e-6 relative: a
            assert (d > 1e-5).mean() < 2e-3, (name, b, d.max() / STEP, (d > 1e-5).mean())     # coordinate may land on the other side of a 1/32 step
        else:
            assert d.max() <= 1e-5, (name, b, d.max() / STEP, (d > 1e-5).mean())


@pytest.mark.parametrize("angle", [-77.0, -12.5, 0.0, 33.0, 45.0, 89.0])
def test_rotate_crop_then_pad_resize(angle):
    from oracle import transforms as T
    rng = np.random.default_rng(int(angle * 10) % 97)
    sizes = [(300, 400), (400, 300), (256, 256), (97, 331)]
    imgs = [_smooth(rng, h, w) for h, w in sizes]
    out = _pipe(224, 224, 4, 400 * 400)(imgs, aug=[dict(angle=angle)] * 4).cpu().numpy()
    for b, img in enumerate(imgs):
        ref = T.train_pipeline(img, 224, 224, dict(angle=angle))
        d = np.abs(out[b] - ref)
        # the rotated image is bit exact; the resize after it may round an x.5 tap sum to the neighbouring uint8 step
        assert d.max() <= STEP + 1e-5 and (d > 1e-5).mean() < 2e-3, (sizes[b], angle, d.max() / STEP, (d > 1e-5).mean())


def test_full_train_pipeline_random_parameters():
    """32 images of different sizes, parameters drawn by the product's own sampler at prob 0.5 (about 2^-8 of the images
    get nothing, most get 3-5 transforms): per image almost every pixel agrees exactly; the few that differ descend from an
    x.5 rounding in the resize, which the later filters can amplify by a few steps."""
    from oracle import transforms as T
    from garbage_classification_rca_amd.preprocess import sample_train_params
    rng = np.random.default_rng(5)
    sizes = [(int(rng.integers(120, 420)), int(rng.integers(120, 420))) for _ in range(32)]
    imgs = [_smooth(rng, h, w) for h, w in sizes]
    params = sample_train_params(np.random.default_rng(9), 32, 0.5)
    pipe = _pipe(224, 224, 32, 420 * 420)
    for rep in range(2):                       # both staging slots
        out = pipe(imgs, aug=params).cpu().numpy()
        frac = []
        for b, img in enumerate(imgs):
            ref = T.train_pipeline(img, 224, 224, params[b])
            d = np.abs(out[b] - ref)
            frac.append((d > 1e-5).mean())
            assert (d > 1e-5).mean() < 1e-2 and d.mean() < 0.02 * STEP, (b, sizes[b], sorted(params[b]), (d > 1e-5).mean(), d.max() / STEP)
        assert np.median(frac) < 1e-3


def test_no_parameters_equals_validation_pipeline():
    from oracle import transforms as T
    rng = np.random.default_rng(2)
    imgs = [_smooth(rng, 200, 310), _smooth(rng, 310, 200)]
    pipe = _pipe(224, 224, 2, 310 * 310)
    a = pipe(imgs, aug=[{}, {}]).cpu()
    b = pipe(imgs).cpu()
    assert torch.equal(a, b)


def test_packed_batches_equal_lists_of_images():
    """main_both.collate_decoded hands the pipeline one packed uint8 tensor per batch (pinned or not)."""
    from garbage_classification_rca_amd.preprocess import pack_images, sample_train_params
    rng = np.random.default_rng(8)
    imgs = [_smooth(rng, int(rng.integers(50, 200)), int(rng.integers(50, 200))) for _ in range(9)]
    params = sample_train_params(np.random.default_rng(3), 9, 0.6)
    pipe = _pipe(128, 96, 16, 200 * 200)
    ref = pipe(imgs, aug=params).cpu()
    packed = pack_images([torch.from_numpy(i) for i in imgs])
    assert packed["flat"].numel() % 16 == 0 and packed["shapes"].tolist() == [list(i.shape[:2]) for i in imgs]
    assert torch.equal(pipe(packed, aug=params).cpu(), ref)
    pinned = {"flat": packed["flat"].pin_memory(), "shapes": packed["shapes"]}
    assert torch.equal(pipe(pinned, aug=params).cpu(), ref)
    assert torch.equal(pipe(pinned).cpu(), pipe(imgs).cpu())
