GENERAL-INFO-START

	seq-file            y9.seq
	trace-file          y9.trace
	locus-mut-rate          CONST
	num-loci            8
	random-seed         12345
	mcmc-iterations	  16
	iterations-per-log  8
	logs-per-line       10

	find-finetunes		FALSE
	finetune-coal-time	0.01		
	finetune-mig-time	0.3		
	finetune-theta		0.04
	finetune-mig-rate	0.02
	finetune-tau		0.0000008
	finetune-mixing		0.003

	tau-theta-print		10000.0
	tau-theta-alpha		1.0
	tau-theta-beta		10000.0

	mig-rate-print		0.001
	mig-rate-alpha		0.002
	mig-rate-beta		0.0000000400

GENERAL-INFO-END

CURRENT-POPS-START	

	POP-START
		name		A
		samples		s0 d
	POP-END

	POP-START
		name		B
		samples		s1 d
	POP-END

	POP-START
		name		C
		samples		s2 d
	POP-END

	POP-START
		name		D
		samples		s3 d
	POP-END

	POP-START
		name		E
		samples		s4 d
	POP-END

	POP-START
		name		F
		samples		s5 d
	POP-END

	POP-START
		name		G
		samples		s6 d
	POP-END

	POP-START
		name		H
		samples		s7 d
	POP-END

	POP-START
		name		I
		samples		s8 d
	POP-END

	POP-START
		name		J
		samples		s9 d
	POP-END

	POP-START
		name		K
		samples		s10 d
	POP-END

	POP-START
		name		L
		samples		s11 d
	POP-END

	POP-START
		name		M
		samples		s12 d
	POP-END

	POP-START
		name		N
		samples		s13 d
	POP-END

	POP-START
		name		O
		samples		s14 d
	POP-END

	POP-START
		name		P
		samples		s15 d
	POP-END

	POP-START
		name		Q
		samples		s16 d
	POP-END

	POP-START
		name		R
		samples		s17 d
	POP-END

	POP-START
		name		S
		samples		s18 d
	POP-END

	POP-START
		name		T
		samples		s19 d
	POP-END

CURRENT-POPS-END

ANCESTRAL-POPS-START

	POP-START
		name			AB
		children		A		B
		tau-initial	0.000005000
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABC
		children		AB		C
		tau-initial	0.000006250
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCD
		children		ABC		D
		tau-initial	0.000007813
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDE
		children		ABCD		E
		tau-initial	0.000009766
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEF
		children		ABCDE		F
		tau-initial	0.000012207
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFG
		children		ABCDEF		G
		tau-initial	0.000015259
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGH
		children		ABCDEFG		H
		tau-initial	0.000019073
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHI
		children		ABCDEFGH		I
		tau-initial	0.000023842
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJ
		children		ABCDEFGHI		J
		tau-initial	0.000029802
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJK
		children		ABCDEFGHIJ		K
		tau-initial	0.000037253
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJKL
		children		ABCDEFGHIJK		L
		tau-initial	0.000046566
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJKLM
		children		ABCDEFGHIJKL		M
		tau-initial	0.000058208
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJKLMN
		children		ABCDEFGHIJKLM		N
		tau-initial	0.000072760
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJKLMNO
		children		ABCDEFGHIJKLMN		O
		tau-initial	0.000090949
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJKLMNOP
		children		ABCDEFGHIJKLMNO		P
		tau-initial	0.000113687
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJKLMNOPQ
		children		ABCDEFGHIJKLMNOP		Q
		tau-initial	0.000142109
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJKLMNOPQR
		children		ABCDEFGHIJKLMNOPQ		R
		tau-initial	0.000177636
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJKLMNOPQRS
		children		ABCDEFGHIJKLMNOPQR		S
		tau-initial	0.000222045
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			root
		children		ABCDEFGHIJKLMNOPQRS		T
		tau-initial	0.001110223
		tau-beta		20000.0	
		finetune-tau			0.00000286
	POP-END

ANCESTRAL-POPS-END

MIG-BANDS-START	
	BAND-START		
       source  A
       target  B
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  B
       target  C
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  C
       target  D
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  D
       target  E
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  E
       target  F
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  F
       target  G
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  G
       target  H
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  H
       target  I
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  B
       target  A
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  C
       target  B
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  D
       target  C
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  E
       target  D
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  F
       target  E
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  G
       target  F
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  H
       target  G
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  I
       target  H
       mig-rate-print 0.1
	BAND-END

MIG-BANDS-END
