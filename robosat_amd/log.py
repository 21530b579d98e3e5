"""Run log for ``rs train``: the interface of the reference's ``robosat.log.Log`` (``Log(path, out).log(msg)``,
``robosat/log.py``), i.e. what ``train.py`` writes to ``<checkpoint>/log`` and echoes to the console."""

import os
import sys


class Log:
    """Line-oriented log.  Each message becomes one line appended to ``path`` (line-buffered, so a killed run keeps
    everything it reported) and, when ``out`` is a stream, one line on that stream as well."""

    def __init__(self, path, out=sys.stdout):
        self._sink = open(path, "a", buffering=1)
        self._echo = out

    def log(self, msg):
        line = str(msg)
        self._sink.write(line + os.linesep)
        if self._echo is not None:
            self._echo.write(line + "\n")
            self._echo.flush()

    def close(self):
        if not self._sink.closed:
            self._sink.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
