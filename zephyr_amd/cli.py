"""Command line: `python -m zephyr_amd model <projnm> [--job OmegaJob]`  (zephyr/frontend/cli.py:70-83).

Only `model` does work in the reference; its other sub-commands print a banner and are kept the same way."""
import argparse
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog='zephyr_amd', description='frequency-domain Helmholtz modelling on MI355X')
    sub = ap.add_subparsers(dest='command')
    for name, text in (('init', 'Set up a new modelling or inversion project'), ('invert', 'Run an inversion project'),
                       ('inspect', 'Print information about an existing project'), ('migrate', 'Run a migration'),
                       ('clean', 'Clean up project results / outputs'), ('pack', 'Collect configuration into an HDF5 datafile'),
                       ('unpack', 'Extract configuration from an HDF5 datafile')):
        p = sub.add_parser(name, help=text)
        p.add_argument('projnm')
    p = sub.add_parser('model', help='Run a forward model')
    p.add_argument('projnm')
    p.add_argument('--job', default='OmegaJob', help='The job to run')
    args = ap.parse_args(argv)
    if args.command is None:
        ap.print_help()
        return 2
    if args.command != 'model':
        print('%s: not implemented (a stub in the reference as well)' % args.command)
        print('projnm: \t%s' % args.projnm)
        return 0
    from . import jobs
    jClass = getattr(jobs, args.job, None)
    if jClass is None or not (isinstance(jClass, type) and issubclass(jClass, jobs.Job)):
        print('unknown job %r' % args.job, file=sys.stderr)
        return 2
    jClass(args.projnm).run()
    return 0
