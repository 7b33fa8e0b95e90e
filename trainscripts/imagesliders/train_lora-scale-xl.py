"""Entry point kept at the reference's path (trainscripts/imagesliders/train_lora-scale-xl.py): SDXL image sliders on the
MI355X engine.  See sliders_amd/cli_image.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sliders_amd.cli_image import main  # noqa: E402

if __name__ == "__main__":
    main(xl=True)
