"""Name-only stub (see ../__init__.py).  model_util.py:11-16 imports these names."""


class DDIMScheduler:
    pass


class DDPMScheduler:
    pass


class LMSDiscreteScheduler:
    pass


class EulerAncestralDiscreteScheduler:
    pass
