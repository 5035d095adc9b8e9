"""envs/__init__.py (drop-in for the reference's envs/__init__.py:1-5): registration with gym is executed, tolerated when
gym is absent or the id is already registered, and LOUD for any other failure."""
import os
import subprocess
import sys
import textwrap

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "atc-reinforcement-learning_amd")


def _run(code, extra_path=()):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([PKG] + list(extra_path))
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=120)


def test_registers_with_the_stand_in_gym():
    r = _run("""
        import envs
        from gym.envs.registration import registry
        assert registry['AtcEnv-v0'] == 'envs.atc.atc_gym:AtcGym'
    """, [os.path.join(HERE, "oracle_shims")])
    assert r.returncode == 0, r.stderr


def test_import_without_gym_is_silent():
    r = _run("import envs")
    assert r.returncode == 0, r.stderr


def test_broken_registration_is_not_swallowed(tmp_path):
    pkg = tmp_path / "gym" / "envs"
    pkg.mkdir(parents=True)
    (tmp_path / "gym" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "registration.py").write_text("def register(id, entry_point, **kw):\n    raise RuntimeError('registry is broken')\n")
    r = _run("import envs", [str(tmp_path)])
    assert r.returncode != 0 and "registry is broken" in r.stderr


def test_second_registration_of_the_same_id_is_tolerated(tmp_path):
    pkg = tmp_path / "gym" / "envs"
    pkg.mkdir(parents=True)
    (tmp_path / "gym" / "__init__.py").write_text("from . import error\n")
    (tmp_path / "gym" / "error.py").write_text("class Error(Exception):\n    pass\n")
    (pkg / "__init__.py").write_text("")
    (pkg / "registration.py").write_text(
        "from gym import error\nregistry = {}\n"
        "def register(id, entry_point, **kw):\n"
        "    if id in registry:\n        raise error.Error('Cannot re-register id: ' + id)\n"
        "    registry[id] = entry_point\n")
    r = _run("import envs, importlib\nimportlib.reload(envs)", [str(tmp_path)])
    assert r.returncode == 0, r.stderr


def test_installed_but_broken_gym_falls_through_to_gymnasium(tmp_path):
    """an installed gym that cannot be imported (old gym on a new NumPy raises AttributeError at import) must not
    make `import envs` fail, and must not prevent the gymnasium fallback."""
    bad = tmp_path / "gym"
    bad.mkdir()
    (bad / "__init__.py").write_text("raise AttributeError(\"module 'numpy' has no attribute 'bool'\")\n")
    good = tmp_path / "gymnasium" / "envs"
    good.mkdir(parents=True)
    (tmp_path / "gymnasium" / "__init__.py").write_text("")
    (good / "__init__.py").write_text("")
    (good / "registration.py").write_text("registry = {}\ndef register(id, entry_point, **kw):\n    registry[id] = entry_point\n")
    r = _run("""
        import warnings
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            import envs
        assert any("failed to import" in str(x.message) for x in w), [str(x.message) for x in w]
        from gymnasium.envs.registration import registry
        assert registry['AtcEnv-v0'] == 'envs.atc.atc_gym:AtcGym'
    """, [str(tmp_path)])
    assert r.returncode == 0, r.stderr
