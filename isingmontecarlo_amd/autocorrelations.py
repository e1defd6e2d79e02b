"""Autocorrelations of sampled observables (qmc::sse::autocorrelations, src/sse/autocorrelations.rs).

Post-processing of the p=0 states a batch samples.  The transform itself runs on the host (numpy FFT; the reference uses
rustfft) or, with `device="cuda"`, on the GPU through hipFFT (torch.fft on a ROCm build dispatches to hipFFT): the batched
form — every variable of every replica is one series — is what a 1024-replica batch needs."""
import numpy as np


def fft_autocorrelation_device(samples, device="cuda"):
    """The same as fft_autocorrelation for a whole batch at once on the GPU (hipFFT): `samples` is [T][R][n], returns [R][T].
    (A process that uses torch's HIP runtime next to this library must let torch touch the GPU first — `torch.cuda.is_available()`
    before the first batch is created, as bench.py does; the other order leaves torch without devices.)"""
    import torch
    x = torch.as_tensor(np.ascontiguousarray(samples), dtype=torch.float64, device=device)
    tmax, _, n = x.shape
    x = x - x.mean(dim=0, keepdim=True)
    norm = torch.sqrt((x * x).sum(dim=0, keepdim=True))
    x = torch.where(norm > 0, x / torch.where(norm > 0, norm, torch.ones_like(norm)), torch.zeros_like(x))
    f = torch.fft.fft(x, dim=0)
    ac = torch.fft.ifft(f * torch.conj(f), dim=0).real * tmax  # (rustfft's inverse is unnormalised)
    return (ac.sum(dim=2) / (n * tmax)).transpose(0, 1).contiguous().cpu().numpy()


def direct_autocorrelation(samples):
    """The defining circular sum, O(T^2) per observable: r[tau] = mean over observables of sum_t y[t] y[(t + tau) mod T] with y
    the centred, unit-norm series (test reference for the FFT forms)."""
    x = np.asarray(samples, dtype=np.float64)
    tmax, n = x.shape
    x = x - x.mean(axis=0, keepdims=True)
    norm = np.sqrt((x * x).sum(axis=0, keepdims=True))
    x = np.divide(x, norm, out=np.zeros_like(x), where=norm > 0)
    return np.array([(x * np.roll(x, -tau, axis=0)).sum() / n for tau in range(tmax)])


def fft_autocorrelation(samples):
    """fft_autocorrelation (autocorrelations.rs:99-133): `samples` is [T][n] (T samples of n observables).  Every
    observable is centred and scaled to unit norm, its circular autocorrelation is taken through the FFT, and the
    result is averaged over the n observables: returns T values, r[0] = 1 (up to rounding) when every observable varied."""
    x = np.asarray(samples, dtype=np.float64)
    tmax, n = x.shape
    x = x - x.mean(axis=0, keepdims=True)
    norm = np.sqrt((x * x).sum(axis=0, keepdims=True))
    x = np.divide(x, norm, out=np.zeros_like(x), where=norm > 0)  # an observable that never changed contributes 0 (the reference divides by 0 there)
    f = np.fft.fft(x, axis=0)
    ac = np.fft.ifft(f * np.conj(f), axis=0).real * tmax  # rustfft's inverse is unnormalised
    return ac.sum(axis=1) / (n * tmax)


def variable_autocorrelation(graph, timesteps, beta, sampling_freq=1, r=None, device=None):
    """QmcAutoCorrelations::calculate_variable_autocorrelation (autocorrelations.rs:37-50) for a batch: runs `timesteps`
    sweeps, samples the p=0 states every `sampling_freq` sweeps and returns the autocorrelation of the +-1 spins, one
    row per replica (or for replica r only).  `device="cuda"`: the transforms of all replicas in one hipFFT batch."""
    states = []
    done = 0
    while done < timesteps:
        t = min(sampling_freq, timesteps - done)
        graph.run(t, beta)
        done += t
        if done % sampling_freq == 0:
            states.append(graph.state_ref())
    st = np.stack(states).astype(np.float64) * 2.0 - 1.0  # [T][R][N]
    if device is not None:
        out = fft_autocorrelation_device(st if r is None else st[:, r:r + 1, :], device)
        return out if r is None else out[0]
    reps = range(st.shape[1]) if r is None else [r]
    out = np.stack([fft_autocorrelation(st[:, k, :]) for k in reps])
    return out if r is None else out[0]


def spin_product_autocorrelation(graph, timesteps, beta, var_products, sampling_freq=1, r=None):
    """calculate_spin_product_autocorrelation (autocorrelations.rs:52-75): observables are products of the +-1 spins of
    each variable group in `var_products`."""
    states = []
    done = 0
    while done < timesteps:
        t = min(sampling_freq, timesteps - done)
        graph.run(t, beta)
        done += t
        if done % sampling_freq == 0:
            states.append(graph.state_ref())
    st = np.stack(states).astype(np.float64) * 2.0 - 1.0
    prods = np.stack([st[:, :, list(vs)].prod(axis=2) for vs in var_products], axis=2)  # [T][R][len(var_products)]
    reps = range(st.shape[1]) if r is None else [r]
    out = np.stack([fft_autocorrelation(prods[:, k, :]) for k in reps])
    return out if r is None else out[0]
