def pytest_configure(config):
    config.addinivalue_line("markers", "experiments: needs a library built from the archived experiment sources")
