"""Deterministic synthetic PCM used by parity tests and bench (SURVEY.md §8d config 2 corpora)."""
import numpy as np

def music(n_frames, frame=960, channels=2, seed=0, amp=12000.0):
    """Harmonic 'music': 11 harmonics of f0 in [80,400] Hz with 5 Hz vibrato, 2 Hz on/off gating (transients),
    white noise at -26 dB; per-channel gains.  int16 interleaved [n_frames*frame, channels]."""
    rng = np.random.default_rng(1234 + seed)
    n = n_frames * frame
    t = np.arange(n) / 48000.0
    f0 = rng.uniform(80, 400)
    ph = 2 * np.pi * f0 * t + (f0 * 0.01 / 5.0) * np.sin(2 * np.pi * 5 * t)
    s = sum(np.sin(k * ph + rng.uniform(0, 6.28)) / k for k in range(1, 12))
    gate = (np.sin(2 * np.pi * 2 * t + rng.uniform(0, 6.28)) > -0.3).astype(np.float64)
    s = s * gate / 2.0
    out = np.zeros((n, channels))
    for c in range(channels):
        out[:, c] = amp * (rng.uniform(.4, 1.0) * s + 0.05 * rng.normal(0, 1, n))
    return np.clip(np.round(out), -32768, 32767).astype(np.int16)

def noise_bursts(n_frames, frame=960, channels=2, seed=0):
    rng = np.random.default_rng(99 + seed)
    n = n_frames * frame
    env = np.repeat(rng.choice([0.0, 0.02, 0.3, 1.0], size=n // 240 + 1), 240)[:n]
    x = rng.normal(0, 6000, (n, channels)) * env[:, None]
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)

def tone(n_frames, frame=960, channels=2, freq=1000.0, amp=20000.0):
    n = n_frames * frame
    t = np.arange(n) / 48000.0
    s = amp * np.sin(2 * np.pi * freq * t)
    return np.clip(np.round(np.stack([s] * channels, 1)), -32768, 32767).astype(np.int16)

def silence_then_music(n_frames, frame=960, channels=2, seed=0):
    x = music(n_frames, frame, channels, seed)
    x[: 3 * frame] = 0
    x[(n_frames // 2) * frame:(n_frames // 2 + 2) * frame] = 0
    return x
