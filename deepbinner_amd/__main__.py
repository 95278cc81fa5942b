from .deepbinner import main

main()
