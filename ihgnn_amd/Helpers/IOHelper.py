"""Tee logging: timestamped lines to stdout and, once started, to a log file without ANSI colour codes
(reference ``Helpers/IOHelper.py:24-77``)."""
import os
import re
import sys
import time

_ANSI = re.compile('\033\\[0;*[0-9]*m')


class IOHelper:
    log_filename = None
    _warned = False
    quiet = False                      # set on non-chief ranks of a multi-process run

    @staticmethod
    def GetAllContent(filename, encoding='utf-8'):
        with open(filename, 'r', encoding=encoding) as f:
            return f.read()

    @staticmethod
    def GetFirstLineContent(filename, encoding='utf-8'):
        with open(filename, 'r', encoding=encoding) as f:
            return f.readline().strip()

    @staticmethod
    def CanLogToFile():
        return IOHelper.log_filename is not None

    @staticmethod
    def StartLogging(log_filename=None):
        if log_filename is None:
            IOHelper._warned = True        # explicit "stdout only": no warning later
            return
        folder = os.path.dirname(log_filename)
        if folder:
            os.makedirs(folder, exist_ok=True)
        with open(log_filename, 'w', encoding='utf-8') as f:
            f.write('\n')
        IOHelper.log_filename = log_filename
        IOHelper.LogPrint(f'logging to {log_filename}')

    @staticmethod
    def EndLogging():
        if IOHelper.CanLogToFile():
            IOHelper.LogPrint(f'log closed: {IOHelper.log_filename}')
            IOHelper.log_filename = None

    @staticmethod
    def LogPrint(message_no_endline='', put_time_in_single_line=False):
        if IOHelper.quiet:
            return
        text = message_no_endline
        if text != '':
            body = text.lstrip('\n')
            lead = text[:len(text) - len(body)]
            stamp = time.strftime('[%H:%M:%S] ', time.localtime()) + ('\n' if put_time_in_single_line else '')
            text = lead + stamp + body
        if IOHelper.CanLogToFile():
            with open(IOHelper.log_filename, 'a', encoding='utf-8') as f:
                f.write(_ANSI.sub('', text) + '\n')
        elif not IOHelper._warned:
            print('Warning: IOHelper.StartLogging() has not been called; logging to stdout only.')
            IOHelper._warned = True
        print(text)
        sys.stdout.flush()

    @staticmethod
    def WriteListToFile(items, filename):
        with open(filename, 'w', encoding='utf-8') as f:
            f.writelines(f'{x}\n' for x in items)
        IOHelper.LogPrint(f'{len(items)} lines written to {filename}')

    @staticmethod
    def ReadStringListFromFile(filename, encoding='utf-8'):
        with open(filename, 'r', encoding=encoding) as f:
            return [line.strip() for line in f]
