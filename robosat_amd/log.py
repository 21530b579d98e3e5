"""Append-mode training log that also echoes to a stream (reference ``robosat/log.py`` behaviour)."""

import os
import sys


class Log:
    def __init__(self, path, out=sys.stdout):
        self.out = out
        self.fp = open(path, "a")

    def log(self, msg):
        self.fp.write(msg + os.linesep)
        self.fp.flush()
        if self.out:
            print(msg, file=self.out)
