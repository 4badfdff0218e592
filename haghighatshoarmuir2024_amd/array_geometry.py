"""Far-field array geometries (host side; feeds per-microphone delays to the device path).

Same call surface as the reference's micloc/array_geometry.py (ArrayGeometry :17-61, CircularArray :64-78,
CenterCircularArray :81-94, LinearArray :97-119, Random2DArray :122-131), including its quirks:
`np.linspace(0, 2*pi, n)` keeps the end point, so two microphones of the circular arrays coincide
(SURVEY A.7).  `delays` additionally accepts an array of DoAs (vectorised), which is what removes the
T-calls-per-trial Python loop of snn_beamformer.py:254-256.
"""
import numpy as np

SOUND_SPEED_IN_OPEN_AIR = 340


class ArrayGeometry:
    """Microphones in polar coordinates (r_m, theta_m); plane wave from DoA theta arrives at microphone m
    with delay  -r_m cos(theta_m - theta) / speed."""

    def __init__(self, r_vec, theta_vec, speed=SOUND_SPEED_IN_OPEN_AIR):
        r_vec = np.asarray(r_vec)
        if np.any(r_vec < 0):
            raise ValueError("distances of the elements in `r_vec` should be all positive!")
        self.r_vec = r_vec
        self.theta_vec = np.asarray(theta_vec)
        self.speed = speed

    def delays(self, theta, normalized=True):
        theta = np.asarray(theta, dtype=np.float64)
        if theta.ndim == 0:
            d = -self.r_vec * np.cos(self.theta_vec - theta) / self.speed
            if normalized:
                d = d - np.min(d)
            return d
        # vectorised: result [len(theta), num_mic]; each row is what the scalar call returns
        d = -self.r_vec[None, :] * np.cos(self.theta_vec[None, :] - theta.reshape(-1, 1)) / self.speed
        if normalized:
            d = d - d.min(axis=1, keepdims=True)
        return d

    def __len__(self):
        return len(self.r_vec)


class CircularArray(ArrayGeometry):
    def __init__(self, radius, num_mic, speed=SOUND_SPEED_IN_OPEN_AIR):
        super().__init__(radius * np.ones(num_mic), np.linspace(0, 2 * np.pi, num_mic), speed=speed)


class CenterCircularArray(ArrayGeometry):
    """num_mic - 1 microphones on the ring plus one in the centre (the 7-mic devkit)."""

    def __init__(self, radius, num_mic, speed=SOUND_SPEED_IN_OPEN_AIR):
        ring = num_mic - 1
        r_vec = np.concatenate([radius * np.ones(ring), [0.0]])
        theta_vec = np.concatenate([np.linspace(0, 2 * np.pi, ring), [0.0]])
        super().__init__(r_vec, theta_vec, speed=speed)


class LinearArray(ArrayGeometry):
    def __init__(self, spacing, num_mic, radius, speed=SOUND_SPEED_IN_OPEN_AIR):
        pos = spacing * (np.arange(-num_mic / 2, num_mic / 2) + 0.5)
        theta_vec = np.where(pos < 0, np.pi, 0.0)
        super().__init__(np.abs(pos), theta_vec, speed=speed)
        self.radius = radius


class Random2DArray(ArrayGeometry):
    """Uniform over a disc; draws from the global NumPy stream in the reference's order (r then theta)."""

    def __init__(self, radius, num_mic, speed=SOUND_SPEED_IN_OPEN_AIR):
        r_vec = np.sqrt(np.random.rand(num_mic)) * radius
        theta_vec = np.random.rand(num_mic) * 2 * np.pi
        super().__init__(r_vec, theta_vec, speed=speed)
        self.radius = radius
