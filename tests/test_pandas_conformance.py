"""pandas' own ExtensionArray conformance suites (pandas.tests.extension.base) against
searcharray_amd.SearchArray -- the "pandas ExtensionArray surface" the drop-in must keep.  The
reference runs the same suites (its test/test_extension_array.py); the fixture names are the ones
pandas' base classes require.  Host bookkeeping only, so this runs on the emulated kernels (CPU)."""
import pandas as pd
import pytest
from pandas.tests.extension import base

from searcharray_amd import SearchArray, Terms, TermsDtype
from searcharray_amd import _lib
from tests.emu import emu_api


@pytest.fixture(autouse=True, scope="module")
def _bind_emulated_kernels():
    old = _lib._api
    _lib.use_api(emu_api())
    yield
    _lib.use_api(old)


CORPUS = ["red green green blue", "solo", "pair green", "one two three"] * 25


@pytest.fixture
def dtype():
    return TermsDtype()


@pytest.fixture
def data():
    return SearchArray.index(CORPUS)


@pytest.fixture
def data_missing():
    return SearchArray.index(["", "red green blue"])


@pytest.fixture
def na_cmp():
    return lambda a, b: a == Terms({}) or b == Terms({})


@pytest.fixture
def na_value():
    return Terms({})


@pytest.fixture(params=[True, False])
def as_series(request):
    return request.param


@pytest.fixture(params=[True, False])
def as_frame(request):
    return request.param


@pytest.fixture
def data_repeated(data):
    def gen(count):
        for _ in range(count):
            yield data
    return gen


@pytest.fixture
def invalid_scalar(data):
    return 123


@pytest.fixture(params=[True, False])
def use_numpy(request):
    return request.param


@pytest.fixture
def data_for_sorting():
    # [B, C, A] with A < B < C under Terms.__lt__ (lexical over the sparse tf vectors)
    return SearchArray.index(["alpha mu delta", "alpha alpha ask", "cab cat"])


@pytest.fixture
def data_missing_for_sorting():
    return SearchArray.index(["alpha mu delta", "", "cab cat"])


@pytest.fixture
def data_for_grouping():
    # [B, B, NA, NA, A, A, B, C]
    return SearchArray.index(["alpha mu delta", "alpha mu delta", "", "", "cab cat", "cab cat",
                              "alpha mu delta", "alpha alpha ask"])


@pytest.fixture(params=[lambda x: 1, lambda x: [1] * len(x), lambda x: pd.Series([1] * len(x)), lambda x: x],
                ids=["scalar", "list", "series", "object"])
def groupby_apply_op(request):
    return request.param


@pytest.fixture(params=["data", "data_missing"])
def all_data(request, data, data_missing):
    return data if request.param == "data" else data_missing


@pytest.fixture(params=[None, lambda x: x])
def sort_by_key(request):
    return request.param


@pytest.fixture(params=[True, False])
def box_in_series(request):
    return request.param


@pytest.fixture(params=[True, False])
def as_array(request):
    return request.param


@pytest.fixture(params=["ffill", "bfill"])
def fillna_method(request):
    return request.param


class TestDType(base.BaseDtypeTests):
    pass


class TestInterface(base.BaseInterfaceTests):
    pass


class TestMethods(base.BaseMethodsTests):
    # unique / normalised value_counts are not meaningful on inverted-index rows (the reference
    # stubs the same two, test/test_extension_array.py:151-159)
    def test_value_counts_with_normalize(self, data):
        pass

    def test_unique(self, data):
        pass


class TestConstructors(base.BaseConstructorsTests):
    pass


class TestReshaping(base.BaseReshapingTests):
    pass


class TestGetItem(base.BaseGetitemTests):
    pass


class TestSetItem(base.BaseSetitemTests):
    pass


class TestCasting(base.BaseCastingTests):
    pass


class TestPrinting(base.BasePrintingTests):
    pass


class TestMissing(base.BaseMissingTests):
    pass


class TestGroupby(base.BaseGroupbyTests):
    pass
