"""Re-export of exploremultimodal_amd.synth (synthetic configs / weights / batches) for the oracle-side scripts and
tests.  The generator itself is input generation, not reference arithmetic, and lives in the package so that the
product benchmark never imports anything under oracle/ except for its cpu_baseline leg."""
from exploremultimodal_amd.synth import *  # noqa: F401,F403
from exploremultimodal_amd.synth import _gen, _normal  # noqa: F401
