"""CPU tests of the non-default noise schedulers (sliders_amd/schedulers.py) against the float64 oracle
(oracle/sched_oracle.py) and against properties / external constants of the algorithms.

The reference selects them with train.noise_scheduler (model_util.py:247-274) and drives them through
scheduler.init_noise_sigma / scale_model_input / step (train_util.py:55,156,193).  diffusers is not installed here: the
parity of these classes is to the published algorithms restated by the oracle ("parity unpinned", see its header)."""
import numpy as np
import pytest
import torch

from oracle import sched_oracle as so
from sliders_amd import schedulers as S
from sliders_amd.model_util import create_noise_scheduler

SHAPE = (2, 4, 8, 8)
PRED = ["epsilon", "v_prediction"]


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def _rand(seed):
    g = np.random.default_rng(seed)
    return g.standard_normal(SHAPE), g.standard_normal(SHAPE), g.standard_normal(SHAPE)


def test_external_constants_of_the_beta_schedule():
    """k-diffusion's constants for scaled_linear 0.00085..0.012 / 1000 steps"""
    sig = so.sigmas_train()
    assert abs(sig.max() - 14.6146) < 1e-3 and abs(sig.min() - 0.0292) < 1e-4
    for cls in (S.EulerAncestralDiscreteScheduler, S.LMSDiscreteScheduler):
        sch = cls()
        assert abs(float(sch.init_noise_sigma) - 14.6146) < 1e-3
        assert float(sch.sigmas[-1]) == 0.0 and abs(float(sch.sigmas[-2]) - 0.0292) < 1e-4
        assert sch.timesteps.dtype == torch.float64 and sch.timesteps[0] == 999.0 and sch.timesteps[-1] == 0.0
    d = S.DDPMScheduler()
    assert d.init_noise_sigma == 1.0 and abs(float(d.alphas_cumprod[-1]) - 0.0047) < 1e-4
    assert abs(float(d.alphas_cumprod[0]) - 0.99915) < 1e-6


def test_factory_names_and_rejection():
    assert type(create_noise_scheduler("ddpm")).__name__ == "DDPMScheduler"
    assert type(create_noise_scheduler("LMS")).__name__ == "LMSDiscreteScheduler"
    assert create_noise_scheduler("euler_a", "v_prediction").prediction_type == "v_prediction"
    assert type(create_noise_scheduler("ddim")).__name__ == "DDIMScheduler"
    with pytest.raises(ValueError):
        create_noise_scheduler("dpm++")
    with pytest.raises(ValueError):
        S.DDPMScheduler(prediction_type="sample")


@pytest.mark.parametrize("n", [50, 1000, 20])
def test_timestep_and_sigma_tables(n):
    d = S.DDPMScheduler()
    d.set_timesteps(n)
    assert d.timesteps.tolist() == so.ddpm_timesteps(n).tolist()
    for cls in (S.EulerAncestralDiscreteScheduler, S.LMSDiscreteScheduler):
        sch = cls()
        sch.set_timesteps(n)
        np.testing.assert_allclose(sch.timesteps.numpy(), so.sigma_timesteps(n), rtol=0, atol=1e-9)
        np.testing.assert_allclose(sch.sigmas.numpy(), so.sigma_schedule(n), rtol=2e-5, atol=1e-7)
        assert sch.sigmas.dtype == torch.float32 and len(sch.sigmas) == n + 1


@pytest.mark.parametrize("pred", PRED)
@pytest.mark.parametrize("i", [0, 7, 48, 49])
def test_ddpm_step_matches_oracle(pred, i):
    x, m, nz = _rand(i)
    sch = S.DDPMScheduler(prediction_type=pred)
    sch.set_timesteps(50)
    t = int(sch.timesteps[i])
    got = sch.step(_t(m), sch.timesteps[i], _t(x), noise=_t(nz)).prev_sample
    want = so.ddpm_step(x, m, t, 50, nz, pred)
    np.testing.assert_allclose(got.numpy(), want, rtol=2e-4, atol=2e-5)
    if t == 0:      # no variance term at the last step: the draw must not matter
        again = sch.step(_t(m), sch.timesteps[i], _t(x), noise=_t(nz) * 100).prev_sample
        assert torch.equal(got, again)


@pytest.mark.parametrize("pred", PRED)
@pytest.mark.parametrize("i", [0, 7, 48, 49])
def test_euler_ancestral_step_and_input_scaling_match_oracle(pred, i):
    x, m, nz = _rand(100 + i)
    x = x * so.sigma_schedule(50)[i]
    sch = S.EulerAncestralDiscreteScheduler(prediction_type=pred)
    sch.set_timesteps(50)
    t = sch.timesteps[i]
    np.testing.assert_allclose(sch.scale_model_input(_t(x), t).numpy(), so.scale_model_input(x, i, 50), rtol=1e-5, atol=1e-6)
    got = sch.step(_t(m), t, _t(x), noise=_t(nz)).prev_sample
    np.testing.assert_allclose(got.numpy(), so.euler_a_step(x, m, i, 50, nz, pred), rtol=2e-4, atol=3e-5)
    if i == 49:     # sigma_to = 0: the step lands on the predicted clean sample, no noise is added
        out = sch.step(_t(m), t, _t(x), noise=_t(nz))
        np.testing.assert_allclose(out.prev_sample.numpy(), out.pred_original_sample.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("pred", PRED)
def test_lms_run_matches_oracle_with_exact_coefficients(pred):
    g = np.random.default_rng(7)
    x0 = g.standard_normal(SHAPE) * so.sigma_schedule(50)[0]
    outs = [g.standard_normal(SHAPE) for _ in range(9)]
    sch = S.LMSDiscreteScheduler(prediction_type=pred)
    sch.set_timesteps(50)
    want = so.lms_run(x0, outs, 50, pred)
    x = _t(x0)
    for i, m in enumerate(outs):
        x = sch.step(_t(m), sch.timesteps[i], x).prev_sample
        np.testing.assert_allclose(x.numpy(), want[i], rtol=5e-4, atol=5e-4)
    assert len(sch.derivatives) == 4
    sch.set_timesteps(50)       # what the reference does at the top of every iteration (train_lora_xl.py:164)
    assert sch.derivatives == []


def test_lms_coefficients_quad_vs_exact_integration():
    sch = S.LMSDiscreteScheduler()
    sch.set_timesteps(50)
    sig = so.sigma_schedule(50)
    for i in (0, 1, 2, 3, 10, 49):
        o = min(i + 1, 4)
        got = [sch.get_lms_coefficient(o, i, j) for j in range(o)]
        np.testing.assert_allclose(got, so.lms_coefficients(sig, i, o), rtol=2e-4, atol=1e-6)
        # the basis polynomials sum to one: the coefficients sum to the step length
        assert abs(sum(got) - (sig[i + 1] - sig[i])) < 1e-4 * abs(sig[i + 1] - sig[i]) + 1e-6


def test_first_lms_step_is_the_deterministic_euler_step():
    x, m, _ = _rand(5)
    x = x * 14.6
    lms, eul = S.LMSDiscreteScheduler(), S.EulerAncestralDiscreteScheduler()
    lms.set_timesteps(50), eul.set_timesteps(50)
    a = lms.step(_t(m), lms.timesteps[0], _t(x)).prev_sample
    s0, s1 = float(eul.sigmas[0]), float(eul.sigmas[1])
    np.testing.assert_allclose(a.numpy(), x + m * (s1 - s0), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["ddpm", "euler_a", "lms"])
def test_bf16_latents_keep_their_dtype_and_stay_close_to_fp32(name):
    """the reference's latents are bf16 (train_lora_xl.py weight_dtype): scalar * tensor ops round to bf16 per op"""
    x, m, nz = _rand(11)
    sch = create_noise_scheduler(name)
    sch.set_timesteps(50)
    t = sch.timesteps[3]
    xs = x * float(sch.init_noise_sigma)
    hi = sch.step(_t(m), t, _t(xs), noise=_t(nz)).prev_sample
    sch.set_timesteps(50)
    lo = sch.step(_t(m, torch.bfloat16), t, _t(xs, torch.bfloat16), noise=_t(nz, torch.bfloat16)).prev_sample
    assert lo.dtype == torch.bfloat16 and sch.scale_model_input(_t(xs, torch.bfloat16), t).dtype == torch.bfloat16
    rel = float((lo.float() - hi).norm() / hi.norm())
    assert rel < 2e-2, rel


@pytest.mark.parametrize("name", ["ddpm", "euler_a"])
def test_drawn_noise_is_reproducible_from_the_generator(name):
    x, m, _ = _rand(12)
    sch = create_noise_scheduler(name)
    sch.set_timesteps(50)
    t = sch.timesteps[5]
    a = sch.step(_t(m), t, _t(x), generator=torch.Generator().manual_seed(3)).prev_sample
    b = sch.step(_t(m), t, _t(x), generator=torch.Generator().manual_seed(3)).prev_sample
    c = sch.step(_t(m), t, _t(x), generator=torch.Generator().manual_seed(4)).prev_sample
    assert torch.equal(a, b) and not torch.equal(a, c)


class _StubUNet:
    """a deterministic stand-in for the UNet: a fixed linear map of the input plus a timestep-dependent offset"""

    class _Out:
        def __init__(self, sample):
            self.sample = sample

    def __call__(self, x, timestep, encoder_hidden_states=None, **kw):
        shift = encoder_hidden_states.mean(dim=(1, 2)).view(-1, 1, 1, 1)
        return _StubUNet._Out(0.3 * x.roll(1, dims=1) + 0.01 * float(timestep) / 1000 + shift)


@pytest.mark.parametrize("name", ["ddpm", "euler_a", "lms"])
def test_denoise_loop_equals_the_reference_shaped_loop(name):
    """schedulers.denoise (what SliderTrainer runs for these schedulers) == train_util.diffusion (the reference's loop
    shape, train_util.py:175-196) on the same stub model, noise stream and guidance scale"""
    from sliders_amd import train_util
    unet = _StubUNet()
    ctx = torch.randn(2, 5, 8, generator=torch.Generator().manual_seed(1)) * 0.1
    k, n, gs = 6, 50, 3.0
    a_s, b_s = create_noise_scheduler(name), create_noise_scheduler(name)
    lat0 = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(2)) * a_s.init_noise_sigma

    def predict(model_input, t):
        out = unet(torch.cat([model_input] * 2), t, encoder_hidden_states=ctx).sample
        u, c = out.chunk(2)
        return u + gs * (c - u)

    torch.manual_seed(9)
    got = S.denoise(a_s, predict, lat0, k, n)
    torch.manual_seed(9)
    b_s.set_timesteps(n)
    want = train_util.diffusion(unet, b_s, lat0, ctx, total_timesteps=k, guidance_scale=gs)
    assert torch.equal(got, want)
    assert not torch.equal(got, lat0)


def test_denoise_loop_euler_matches_the_float64_oracle():
    g = np.random.default_rng(3)
    k, n = 5, 50
    x0 = g.standard_normal(SHAPE) * so.sigma_schedule(n)[0]
    noises = [g.standard_normal(SHAPE) for _ in range(k)]
    model = lambda z, i: 0.5 * z + 0.1 * i          # acts on the SCALED input
    x = x0
    for i in range(k):
        x = so.euler_a_step(x, model(so.scale_model_input(x, i, n), i), i, n, noises[i])
    sch = S.EulerAncestralDiscreteScheduler()
    it = iter(noises)
    sch._randn = lambda like, generator: _t(next(it))
    step = [0]

    def predict(model_input, t):
        i = step[0]
        step[0] += 1
        return 0.5 * model_input + 0.1 * i

    got = S.denoise(sch, predict, _t(x0), k, n)
    np.testing.assert_allclose(got.numpy(), x, rtol=5e-4, atol=5e-4)


@pytest.mark.parametrize("pred", PRED)
def test_sdxl_euler_scheduler_tables_and_steps(pred):
    """EulerDiscreteScheduler as the SDXL checkpoints configure it (leading spacing, steps_offset 1)"""
    n = 50
    sch = S.EulerDiscreteScheduler(prediction_type=pred)
    sch.set_timesteps(n)
    assert sch.timesteps[0] == 981.0 and sch.timesteps[-1] == 1.0 and len(sch.timesteps) == n
    sig = so.leading_sigma_schedule(n)
    np.testing.assert_allclose(sch.sigmas.numpy(), sig, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(float(sch.init_noise_sigma), np.sqrt(sig.max() ** 2 + 1), rtol=1e-5)
    x, m, _ = _rand(21)
    x = x * float(sch.init_noise_sigma)
    for i in (0, 10, 49):
        got = sch.step(_t(m), sch.timesteps[i], _t(x)).prev_sample
        np.testing.assert_allclose(got.numpy(), so.euler_step(x, m, i, sig, pred), rtol=2e-4, atol=3e-5)
    # epsilon prediction: the step is x + eps * (sigma_next - sigma), so a full run with eps = x / sigma_0 scales x by sigma/sigma_0
    if pred == "epsilon":
        y = _t(x)
        for i in range(n):
            y = sch.step(y / sch.sigmas[i], sch.timesteps[i], y).prev_sample
        assert float(y.abs().max()) < 1e-4
    with pytest.raises(ValueError):
        S.EulerDiscreteScheduler(timestep_spacing="trailing")
