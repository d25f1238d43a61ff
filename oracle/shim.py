"""Import shim that makes the reference (/root/reference, read-only) importable in the BUILD container.

Used only by oracle/gen_golden.py to generate tests/golden/*.npz and to validate oracle/ref_cpu.py;
never on the GPU box (the reference does not exist there) and never by the product package.

Why a shim: the reference imports `wandb`, `deepspeed`, three names removed from
`transformers.generation` after 4.31 and the unused MPT family (whose converter imports private
helpers that transformers 5.x no longer has).  The stubs below provide exactly those names; no
reference source is modified or copied.
"""
import contextlib
import sys
import types

REFERENCE_ROOT = "/root/reference"


def install():
    sys.dont_write_bytecode = True          # never leave __pycache__ inside the reference mount
    import transformers  # noqa: F401  (must be imported BEFORE the deepspeed stub: it probes find_spec)
    import transformers.generation
    import transformers.generation.utils as gu
    from transformers import CLIPVisionModel, LlamaForCausalLM, LlamaModel  # noqa: F401

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "wandb" not in sys.modules:
        stub("wandb", run=None, log=lambda *a, **k: None)
    if "deepspeed" not in sys.modules:
        ds = stub("deepspeed")
        ds.zero = stub("deepspeed.zero", Init=contextlib.nullcontext, GatheredParameters=contextlib.nullcontext)
        ds.comm = stub("deepspeed.comm", barrier=lambda *a, **k: None)
    stub("llava.model.language_model.llava_mpt", LlavaMPTForCausalLM=None, LlavaMPTConfig=None)
    for name in ("SampleOutput", "SampleEncoderDecoderOutput"):
        if not hasattr(gu, name):
            setattr(gu, name, object)
    if not hasattr(transformers.generation, "validate_stopping_criteria"):
        transformers.generation.validate_stopping_criteria = lambda sc, ml: sc
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
