"""Small generators shared by tests."""
import numpy as np


def stretching(N, theta_s=6.0, theta_b=0.3):
    """Song & Haidvogel (1994) Cs_r at rho points (what a ROMS file stores as Cs_r)."""
    s = -1.0 + (np.arange(N) + 0.5) / N
    return (1 - theta_b) * np.sinh(theta_s * s) / np.sinh(theta_s) + theta_b * (np.tanh(theta_s * (s + 0.5)) / (2 * np.tanh(0.5 * theta_s)) - 0.5)
