"""Case-preserving ini reader (reference: utils/read_config.py:15-18); the .ini files of the reference
(sections [User] [Network] [STFT] [Training] [DataFrame]) parse unchanged."""
from configparser import ConfigParser


class myconf(ConfigParser):
    def __init__(self, defaults=None):
        super().__init__(defaults=None)

    def optionxform(self, optionstr):
        return optionstr
