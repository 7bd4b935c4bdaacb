"""Action / observation space descriptors — mirror src/Gym/Spaces/{Space,Box,Discrete}.cs.

Single samples are drawn on the host with numpy's RandomState (the reference uses NumSharp's
np.random, Box.cs:33, Discrete.cs:16).  Batched sampling for a whole VectorEnv runs on the GPU:
VectorEnv.SampleActions() -> gymnet_vecenv_sample_actions (Philox, see csrc/kernels.hip).
"""
import enum

import numpy as np


class Space:                                                    # Space.cs:5-18
    def __init__(self, shape, dtype):
        self.Shape = tuple(shape) if shape is not None else None
        self.DType = np.dtype(dtype)

    def Sample(self, mask=None):
        raise NotImplementedError

    def Contains(self, x):
        raise NotImplementedError

    def Seed(self, seed):
        raise NotImplementedError


class Discrete(Space):                                          # Discrete.cs:5-44
    def __init__(self, n, dtype=np.float32, seed=-1, start=0, random_state=None):
        super().__init__((n,), dtype)
        self.N = int(n)
        self.Start = int(start)
        self.RandomState = np.random.RandomState(seed) if seed != -1 else (random_state or np.random)

    def Sample(self, mask=None):                                # Discrete.cs:17-28
        if mask is not None:
            valid = np.nonzero(np.asarray(mask) == 1)[0]
            if valid.size:
                return self.Start + int(self.RandomState.choice(valid))
            return self.Start
        return self.Start + int(self.RandomState.randint(0, self.N))

    def Contains(self, x):                                      # Discrete.cs:30-44: 0 <= x < N (ignores Start)
        if isinstance(x, enum.Enum):                            # Contains(Enum x), Discrete.cs:42-44
            x = int(x.value)
        if isinstance(x, (bool, np.bool_)) or not isinstance(x, (int, np.integer)):
            raise NotImplementedError(str(x))                   # NotSupportedException
        return 0 <= int(x) < self.N

    def Seed(self, seed):
        self.RandomState = np.random.RandomState(seed)

    def __repr__(self):
        return f"Discrete({self.N})"


class Box(Space):                                               # Box.cs:15-96
    BOTH, BELOW, ABOVE = "Both", "Below", "Above"               # BoundedMannerEnum, Box.cs:9-14

    def __init__(self, low, high, shape=None, dtype=np.float32, seed=-1, random_state=None):
        dtype = np.dtype(dtype)
        if np.isscalar(low) and np.isscalar(high):              # Box.cs:25-34
            if shape is None:
                shape = ()
            low = np.full(shape, low, dtype=dtype)
            high = np.full(shape, high, dtype=dtype)
        else:                                                   # Box.cs:37-47
            if low is None:
                raise ValueError("low")                         # ArgumentNullException
            if high is None:
                raise ValueError("high")
            low = np.asarray(low).astype(dtype)
            high = np.asarray(high).astype(dtype)
            assert low.shape == high.shape
            shape = low.shape
        super().__init__(shape, dtype)
        self.Low, self.High = low, high
        self.RandomState = np.random.RandomState(seed) if seed != -1 else (random_state or np.random)
        self.BoundedLow = self.Low > -np.inf                    # CheckBounded, Box.cs:49-54
        self.BoundedHigh = self.High < np.inf

    def IsBounded(self, manner=BOTH):                           # Box.cs:56-70
        below, above = bool(np.all(self.BoundedLow)), bool(np.all(self.BoundedHigh))
        if manner == Box.BOTH:
            return below and above
        if manner == Box.ABOVE:
            return above
        if manner == Box.BELOW:
            return below
        raise ValueError("Unsupported BoundedMannerEnum value.")

    def Sample(self, mask=None):                                # Box.cs:72-93 — the reference's four regimes
        if mask is not None:
            raise NotImplementedError("Box.sample cannot be provided a mask.")
        unbounded = ~self.BoundedLow & ~self.BoundedHigh
        upp_bounded = ~self.BoundedLow & self.BoundedHigh
        low_bounded = self.BoundedLow & ~self.BoundedHigh
        bounded = self.BoundedLow & self.BoundedHigh
        sample = np.empty(self.Shape, dtype=np.float64)
        rs = self.RandomState
        sample[unbounded] = rs.normal(0.5, 1.0, int(unbounded.sum()))                                 # Box.cs:82 (sic)
        sample[low_bounded] = rs.exponential(1.0, int(low_bounded.sum())) + self.Low[low_bounded]     # Box.cs:83
        sample[upp_bounded] = rs.exponential(1.0, int(upp_bounded.sum())) + self.High[upp_bounded]    # Box.cs:84 (sic)
        sample[bounded] = rs.uniform(self.Low[bounded], self.High[bounded])                           # Box.cs:85
        # Box.cs:86-89: np.floor for EXACTLY int32 / uint32 / byte; any other integer dtype goes straight to astype, which truncates
        # toward zero (floor(-0.5) = -1 but (long) -0.5 = 0) — mirrored, quirk included
        if self.DType in (np.dtype(np.int32), np.dtype(np.uint32), np.dtype(np.uint8)):
            sample = np.floor(sample)
        return sample.astype(self.DType)

    def Contains(self, x):                                      # Box.cs:95-99
        if not isinstance(x, np.ndarray):
            raise NotImplementedError(str(x))
        return x.shape == self.Shape and bool(np.all(x >= self.Low)) and bool(np.all(x <= self.High))

    def Seed(self, seed):
        self.RandomState = np.random.RandomState(seed)

    def __eq__(self, other):                                    # Box.cs:116-120
        return isinstance(other, Box) and np.array_equal(self.Low, other.Low) and np.array_equal(self.High, other.High)

    def __hash__(self):
        return hash((self.Low.tobytes(), self.High.tobytes()))

    def __repr__(self):
        return f"Box{self.Shape}"
