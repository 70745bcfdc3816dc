import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.hookimpl(trylast=True)  # after the faulthandler plugin's handler
def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU")
    # A SIGABRT of any library (the HSA runtime after a GPU fault, glibc's
    # heap checks) kills the whole run; Python's faulthandler can only say
    # where the Python threads were. Have the aborting thread's native
    # backtrace written to the terminal's stderr (a copy made before the
    # capture of the tests starts) first.
    try:
        import oracle_lib
        oracle_lib.build()
        fd = os.dup(sys.__stderr__.fileno())
        oracle_lib.lib().cmio_install_abort_backtrace(fd)
    except Exception as err:  # diagnosis only: never in the way of a run
        print("conftest: no abort backtrace (%s)" % err, file=sys.stderr)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


# Order of the files in a run: the core parity rows of SURVEY.md section 8(a)
# first (transport, fixtures, physics), then the full-size property runs,
# then decomposition / groups, then the f rows (sources, library mode, I/O,
# trackers, executable). With `-x`, whatever breaks must not hide a3-a23.
FILE_ORDER = [
    "test_c_abi.py",
    "test_oracle_pinning.py",
    "test_oracle_physics.py",
    "test_oracle_reemission_branches.py",
    "test_oracle_subgrid.py",
    "test_oracle_emissivity.py",
    "test_gpu_transport.py",
    "test_gpu_fixtures.py",
    "test_gpu_physics.py",
    "test_gpu_fullsize.py",
    "test_gpu_fullsize_physics.py",
    "test_gpu_domain.py",
    "test_domain_distributed.py",
    "test_replica_distributed.py",
    "test_gpu_bench_ranks.py",
    "test_continuous_sources.py",
    "test_cmi_library.py",
    "test_gpu_emissivity.py",
    "test_gpu_trackers.py",
    "test_hdf5_writer.py",
    "test_hdf5_reader.py",
    "test_host_driver.py",
]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(FILE_ORDER)}

    def key(item):
        name = os.path.basename(str(item.fspath))
        return rank.get(name, len(FILE_ORDER))

    items.sort(key=key)  # stable: the order inside a file is kept
