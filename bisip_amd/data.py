"""Bundled example spectra, looked up by name -- the role of the reference's ``DataFiles``
(src/bisip/data.py:13-19): ``DataFiles()['SIP-K389175']`` is the path of that data file."""

from pathlib import Path

_DATA_DIR = Path(__file__).resolve().parent / 'data'


class DataFiles(dict):
    """``{file stem: absolute path}`` of every ``*.dat`` spectrum shipped with the package;
    extra entries may be passed like to ``dict``."""

    def __init__(self, *args, **kwargs):
        bundled = {p.stem: str(p) for p in sorted(_DATA_DIR.glob('*.dat'))}
        super().__init__(bundled)
        self.update(*args, **kwargs)
