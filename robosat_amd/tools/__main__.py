"""``rs`` entry point for the hot-path tools and their immediate callers: ``rs train``,
``rs predict``, ``rs weights``, ``rs masks``, ``rs serve``.  The reference's ``robosat/tools/__main__.py`` registers 15
tools; the other 10 are dataset preparation / vector post-processing and out of scope here."""

import argparse

from robosat_amd.tools import masks, predict, serve, train, weights


def add_parsers():
    parser = argparse.ArgumentParser(prog="./rs")
    subparser = parser.add_subparsers(title="robosat tools", metavar="")
    train.add_parser(subparser)
    predict.add_parser(subparser)
    weights.add_parser(subparser)
    masks.add_parser(subparser)
    serve.add_parser(subparser)
    subparser.required = True
    return parser.parse_args()


def main():
    args = add_parsers()
    args.func(args)


if __name__ == "__main__":
    main()
