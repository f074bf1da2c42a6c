"""Nested dict -> nested argparse.Namespace (same contract as reference Arg_Parser.py:3-12)."""
from argparse import Namespace


def Recursive_Parse(args_Dict):
    return Namespace(**{key: Recursive_Parse(value) if isinstance(value, dict) else value
                        for key, value in args_Dict.items()})


def Load_Hyper_Parameters(path):
    """YAML file -> Namespace.  Reference-written checkpoints dump python-object YAML (Train.py:317-320),
    which needs the full Loader; plain hyper-parameter files load with either."""
    import yaml
    with open(path, encoding="utf-8") as f:
        return Recursive_Parse(yaml.load(f, Loader=yaml.Loader))
