"""Entry point kept at the reference's path (trainscripts/textsliders/train_lora_xl.py): SDXL text sliders on the
MI355X engine.  See sliders_amd/cli.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sliders_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main(xl=True)
