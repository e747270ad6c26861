import re, sys
for l in open(sys.argv[1]):
    if l.startswith('=='):
        print('\n' + l.strip(), end=': ')
    m = re.search(r'"shape": "(\w+)".*"us": ([\d.]+)', l)
    if m:
        print(m.group(1), m.group(2), end='  ')
print()
