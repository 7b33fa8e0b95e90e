"""Entry point kept at the reference's path (trainscripts/textsliders/train_lora.py): SD-1.x text sliders on the
MI355X engine.  See sliders_amd/cli.py (SD-1.x needs the head_dim 40/80/160 attention variants: DESIGN.md 7)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sliders_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main(xl=False)
