class Error(Exception):
    pass


class UnregisteredEnv(Error):
    pass
