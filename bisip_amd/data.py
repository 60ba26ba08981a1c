"""Bundled example spectra (reference: src/bisip/data.py:13-19)."""

import glob
import os


class DataFiles(dict):
    """Maps a data-file name (without extension) to its absolute path."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', '*.dat')
        for path in sorted(glob.glob(here)):
            self[os.path.splitext(os.path.basename(path))[0]] = path
