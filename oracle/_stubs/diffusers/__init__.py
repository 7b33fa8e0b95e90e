"""Name-only stub of the `diffusers` package (NOT installed in this image).

The reference's trainscripts/textsliders/{lora,train_util,model_util}.py import diffusers
symbols for type annotations and for model loading.  Putting this directory on sys.path lets
the reference's own, unmodified Python files be imported in this container so that
tests/golden/make_golden.py can run the reference's LoRANetwork / predict_noise / diffusion /
PromptEmbedsPair.loss over the oracle UNet and pin the oracle.  The classes carry no
behaviour.  Test infrastructure only.
"""


class UNet2DConditionModel:  # annotation only
    pass


class SchedulerMixin:  # annotation only
    pass


class StableDiffusionPipeline:  # name only (model_util.py:5-10)
    pass


class StableDiffusionXLPipeline:  # name only
    pass


class AutoencoderKL:  # name only (imagesliders/model_util.py)
    pass
