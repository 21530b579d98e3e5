"""``rs`` entry point for the hot-path tools: ``rs train`` and ``rs predict`` (reference ``robosat/tools/__main__.py``
registers 15 tools; the other 13 are dataset preparation / vector post-processing and are out of scope here)."""

import argparse

from robosat_amd.tools import predict, train


def add_parsers():
    parser = argparse.ArgumentParser(prog="./rs")
    subparser = parser.add_subparsers(title="robosat tools", metavar="")
    train.add_parser(subparser)
    predict.add_parser(subparser)
    subparser.required = True
    return parser.parse_args()


def main():
    args = add_parsers()
    args.func(args)


if __name__ == "__main__":
    main()
