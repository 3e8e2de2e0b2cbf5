"""The text reader of gadf_add_dataset(path) (gadfit_amd/csrc/reader.cpp): host code, no GPU.  Checked against
(1) hand-written expectations for the syntax list-directed input accepts, (2) Fortran's own list-directed input -- the reference's
two passes restated in tests/fortran/list_directed_reader.F90 -- on the same files, (3) itself on one thread and on several."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from gadfit_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLANG = shutil.which('amdflang') or '/opt/rocm/bin/amdflang'

SAMPLE = ('# x y sigma\n'
          '1.0 2.0 0.1\n'
          'foo bar\n'
          '\n'
          '2.0, 3.5, 0.2\n'
          '\t3.0\t1.25\t0.3 trailing text\n'
          '  4.0d0  -1.5D-1  2.5e-1\n'
          '+5. .5 1,extra\n'
          '6 7 8 9 10\r\n'
          '   \n'
          'x 1 2 3\n'
          '7e0,8E+0,9e-0\n')
WANT = ([1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0], [2.0, 3.5, 1.25, -0.15, 0.5, 7.0, 8.0], [0.1, 0.2, 0.3, 0.25, 1.0, 8.0, 9.0])


def test_syntax_of_records(tmp_path):
    f = tmp_path / 'd.txt'
    f.write_text(SAMPLE, newline='')
    x, y, w = _lib.read_columns(str(f), 3)
    assert x.tolist() == WANT[0] and y.tolist() == WANT[1] and w.tolist() == WANT[2]
    x2, y2 = _lib.read_columns(str(f), 2)
    assert x2.tolist() == WANT[0] and y2.tolist() == WANT[1]
    e3 = tmp_path / 'e3.txt'
    # an exponent introduced by its sign alone: what Fortran's E / ES edit descriptors write once the exponent has three digits
    e3.write_text('9.62390099999999935+195 1.5+3 -2.5-101\n1.-2 .5+0 3+2\n')
    x, y, w = _lib.read_columns(str(e3), 3)
    assert (x.tolist(), y.tolist(), w.tolist()) == ([9.62390099999999935e195, 0.01], [1500.0, 0.5], [-2.5e-101, 300.0])
    g = tmp_path / 'rep.txt'
    g.write_text('2*1.5 3.0\n1.0 2*2.5\n')                    # r*c: r copies of c
    x, y, w = _lib.read_columns(str(g), 3)
    assert (x.tolist(), y.tolist(), w.tolist()) == ([1.5, 1.0], [1.5, 2.5], [3.0, 2.5])
    h = tmp_path / 'signs.txt'
    h.write_text('+-1 2 3\n+1 -2 +3\n1 +-2 3\n')                  # two signs are no number: a skipped record, then a malformed one
    with pytest.raises(_lib.GadfitHipError, match=r'line 3'):
        _lib.read_columns(str(h), 3)
    h.write_text('+-1 2 3\n+1 -2 +3\n')
    assert [a.tolist() for a in _lib.read_columns(str(h), 3)] == [[1.0], [-2.0], [3.0]]


def test_a_record_that_begins_with_a_number_must_be_complete(tmp_path):
    f = tmp_path / 'bad.txt'
    f.write_text('1.0 2.0 3.0\n# fine\n4.0 5.0\n6.0 7.0 8.0\n')
    with pytest.raises(_lib.GadfitHipError, match=r'line 3: fewer than 3 numbers'):
        _lib.read_columns(str(f), 3)
    x, y = _lib.read_columns(str(f), 2)                      # two columns: every record is complete
    assert x.tolist() == [1.0, 4.0, 6.0] and y.tolist() == [2.0, 5.0, 7.0]
    g = tmp_path / 'bad2.txt'
    g.write_text('1.0 abc\n')
    with pytest.raises(_lib.GadfitHipError, match='line 1'):
        _lib.read_columns(str(g), 2)
    with pytest.raises(_lib.GadfitHipError, match='Cannot open'):
        _lib.read_columns(str(tmp_path / 'nope.txt'), 2)
    e = tmp_path / 'empty.txt'
    e.write_text('')
    assert _lib.read_columns(str(e), 2)[0].size == 0


def test_pieces_parsed_on_threads_are_the_file_in_order(tmp_path, monkeypatch):
    rng = np.random.default_rng(11)
    n = 40_000
    x = rng.normal(size=n) * 10.0 ** rng.integers(-30, 30, n); y = rng.normal(size=n); s = rng.uniform(0.1, 2.0, n)
    f = tmp_path / 'big.txt'
    with open(f, 'w') as fh:
        fh.write('# header\n')
        for k in range(n):
            fh.write('%.17g %.17g %.17g\n' % (x[k], y[k], s[k]))
            if k % 997 == 0:
                fh.write('comment %d\n\n' % k)
    got = {}
    for threads in ('1', '7', '16'):
        monkeypatch.setenv('GADFIT_HIP_READ_THREADS', threads)
        got[threads] = _lib.read_columns(str(f), 3)
    for t in ('1', '7', '16'):
        assert np.array_equal(got[t][0], x) and np.array_equal(got[t][1], y) and np.array_equal(got[t][2], s)   # %.17g round-trips


@pytest.mark.skipif(not os.path.exists(FLANG), reason='needs amdflang')
@pytest.mark.parametrize('ncol', [2, 3])
def test_against_fortran_list_directed_input(tmp_path, ncol):
    """the same files through the reference's two list-directed passes (compiled here from tests/fortran/list_directed_reader.F90)"""
    exe = tmp_path / 'ldr'
    subprocess.run([FLANG, '-O1', os.path.join(ROOT, 'tests', 'fortran', 'list_directed_reader.F90'), '-o', str(exe)], check=True,
                   capture_output=True, timeout=300)
    f = tmp_path / 'd.txt'
    f.write_text(SAMPLE, newline='')
    gold = os.path.join(ROOT, 'tests', 'golden')
    files = [str(f)] + [os.path.join(gold, n) for n in ('gaussian_xy.txt', 'integral_double_xys.txt', 'piecewise_aux_xys.txt', 'curve1_xy.txt')]
    for path in files:
        if ncol == 3 and path.endswith('_xy.txt'):
            continue                                           # two-column files
        out = subprocess.run([str(exe), path, str(ncol)], capture_output=True, text=True, timeout=120, check=True).stdout.split('\n')
        n = int(out[0])
        ref = np.array([[float(v) for v in ln.split()] for ln in out[1:1 + n]]).reshape(n, 3)
        cols = _lib.read_columns(path, ncol)
        assert cols[0].size == n, path
        assert np.array_equal(cols[0], ref[:, 0]) and np.array_equal(cols[1], ref[:, 1]), path
        if ncol == 3:
            assert np.array_equal(cols[2], ref[:, 2]), path


def _number(rng):
    v = float(rng.choice([rng.uniform(-1e3, 1e3), rng.uniform(-1, 1), float(rng.integers(-50, 50)), rng.uniform(-1, 1) * 10.0 ** int(rng.integers(-300, 300))]))
    style = int(rng.integers(0, 10))
    big = abs(v) >= 1e9
    if style in (0, 9) and not big:
        return '%d' % int(v) + ('.' if style == 9 else '')
    if style == 1:
        return repr(v)
    if style == 3:
        return '%.6E' % v
    if style in (4, 5):
        return ('%.10e' % v).replace('e', 'd' if style == 4 else 'D')
    if style == 6:                                   # exponent without a letter: 1.5+3
        m, e = ('%.5e' % v).split('e')
        return m + ('%+d' % int(e))
    if style == 7 and not big:
        t = '%.4f' % v
        return t.replace('0.', '.', 1) if abs(v) < 1 else t
    if style == 8 and not big:
        return ('+' if v >= 0 else '') + '%.3f' % v
    return '%.6e' % v


@pytest.mark.skipif(not os.path.exists(FLANG), reason='needs amdflang')
def test_random_files_against_fortran_list_directed_input(tmp_path):
    """60 seeded random files: numbers in every notation list-directed input takes (integers, a bare point, D and E exponents in
    either case, exponents introduced by their sign alone, leading + signs, up to 1e+-300), blanks / tabs / commas between them, records
    with more numbers than asked for, trailing text, comment and blank lines, CR LF -- through the library's reader and through
    Fortran's own list-directed reads (tests/fortran/list_directed_reader.F90): the same records, bit for bit.  (1200 more such files
    were compared when the test was written: HISTORY.md.)"""
    exe = tmp_path / 'ldr'
    subprocess.run([FLANG, '-O1', os.path.join(ROOT, 'tests', 'fortran', 'list_directed_reader.F90'), '-o', str(exe)], check=True,
                   capture_output=True, timeout=300)
    for seed in range(60):
        rng = np.random.default_rng(seed)
        ncol = int(rng.choice([2, 3]))
        lines = []
        for _ in range(int(rng.integers(1, 40))):
            if rng.integers(0, 12) == 0:
                lines.append(str(rng.choice(['# comment', 'x y z', '', '   ', '! 1 2 3', 'abc 1 2 3', '"q" 1 2'])))
                continue
            body = _number(rng)
            for _ in range(ncol + int(rng.integers(0, 3)) - 1):
                body += str(rng.choice([' ', '  ', '\t', ',', ', ', ' ,', ' , ', '   \t '])) + _number(rng)
            lines.append(str(rng.choice(['', ' ', '\t', '   '])) + body + str(rng.choice(['', ' ', ' trailing', ' # note', '\r'])))
        path = tmp_path / ('f%d.txt' % seed)
        with open(path, 'w', newline='') as fh:
            fh.write('\n'.join(lines) + str(rng.choice(['\n', '', '\n\n'])))
        out = subprocess.run([str(exe), str(path), str(ncol)], capture_output=True, text=True, timeout=120, check=True).stdout.split('\n')
        n = int(out[0])
        # (three-digit exponents come back without their letter)
        ref = np.array([[float(re.sub(r'(\d)([+-]\d{3})$', r'\1e\2', v)) for v in ln.split()] for ln in out[1:1 + n]]).reshape(n, 3)
        cols = _lib.read_columns(str(path), ncol)
        assert cols[0].size == n, (seed, cols[0].size, n)
        assert np.array_equal(cols[0], ref[:, 0]) and np.array_equal(cols[1], ref[:, 1]) and (ncol == 2 or np.array_equal(cols[2], ref[:, 2])), seed
