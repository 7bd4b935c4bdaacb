"""Error vocabulary of the path — mirrors src/Gym/Exceptions/*.cs (paths relative to the Gym.NET tree)."""


class GymNetError(RuntimeError):
    """Any failure reported by libgymnet_amd.so that has no reference counterpart (HIP errors ...)."""


class NoDeviceError(GymNetError):
    """No usable AMD GPU.  The engine has no CPU fallback by design."""


class InvalidActionError(Exception):
    """src/Gym/Exceptions/InvalidActionError.cs:7-10"""

    def __init__(self, msg="Action is outside of the configured action space."):
        super().__init__(msg)


class AlreadySteppingError(Exception):
    """src/Gym/Exceptions/AlreadySteppingError.cs:8-10"""

    def __init__(self, msg="already running an async step"):
        super().__init__(msg)


class NotSteppingError(Exception):
    """src/Gym/Exceptions/NotSteppingError.cs:4-6"""

    def __init__(self, msg="not running an async step"):
        super().__init__(msg)
