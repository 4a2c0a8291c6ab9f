"""``python -m bonito_amd <tool> ...`` (same sub-command layout as the reference's bonito/__init__.py:12-32)."""
import sys
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser

from bonito_amd import __version__
from bonito_amd.cli import basecaller

modules = ["basecaller"]


def main(argv=None):
    parser = ArgumentParser("bonito_amd", formatter_class=ArgumentDefaultsHelpFormatter)
    parser.add_argument("-v", "--version", action="version", version="%(prog)s {}".format(__version__))
    sub = parser.add_subparsers(title="subcommands", description="valid commands", dest="command")
    sub.required = True
    for name in modules:
        mod = globals()[name]
        p = sub.add_parser(name, parents=[mod.argparser()])
        p.set_defaults(func=mod.main)
    args = parser.parse_args(argv)
    args._argv = list(sys.argv[1:] if argv is None else argv)[1:]      # the sub-command's own arguments (multi-GPU re-launch)
    return args.func(args)


if __name__ == "__main__":
    sys.exit(main())
